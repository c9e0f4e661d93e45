#!/bin/bash
# round 5, call 9: probe -- the step's appends on a second stream (tools/dbg/two_stream_probe.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c9
export TMPDIR=/tmp
( for c in 131072 32768 4096; do timeout 600 python tools/dbg/two_stream_probe.py $c 4 32 2>&1 | grep -v amdgpu.ids; done ) > ${O}_probe.txt 2>&1
cat ${O}_probe.txt
