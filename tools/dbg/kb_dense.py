import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import kbench
kbench.run(4, 131149, False)
