#!/bin/bash
# developer tool: the strip kernel with its debug switches (HIPSDP_G4_DBG), correctness on the small shapes and time at C2
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do echo "== DBG $d"; HIPSDP_G4_DBG=$d timeout -k 10 120 python3 tests/devtools/gemm4_lab.py small 2>&1 | grep -v "^ALL\|FAIL"; done
for d in 0 4 8 12 16 20 28 32 60; do echo "== DBG $d (timing only)"; HIPSDP_G4_DBG=$d timeout -k 10 120 python3 tests/devtools/gemm4_lab.py c2 2>&1 | grep -v "^ALL\|FAIL"; done
