#!/bin/bash
# developer helper: build from the repo root, then run a command on the GPU box.  usage: tools/gb.sh [timeout] '<command>'
cd /root/repo || exit 1
make -C scip-sdp_amd -j8 2>&1 | grep -E "error|Error|Stop" && exit 1
T=600
if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
/usr/local/graft/bin/gpurun --timeout $T -- "$1" 2>&1 | grep -v "^\[gpurun\] sending"
