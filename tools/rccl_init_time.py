#!/usr/bin/env python3
"""rccl_init_time.py - developer tool: how long the first RCCL bootstrap of a process takes on this box (ncclGetUniqueId +
ncclCommInitRank of a one-rank communicator through libhipsdp.so), with the environment as given.  Usage:
   NCCL_DEBUG=INFO python tools/rccl_init_time.py        (add NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 to compare)"""
import ctypes as C, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
lib = hb.lib()
t0 = time.time()
uid = (C.c_ubyte * 128)()
assert lib.hipsdp_comm_unique_id(uid) == 0
t1 = time.time()
comm = C.c_void_p()
assert lib.hipsdp_comm_create(uid, 0, 1, C.byref(comm)) == 0
t2 = time.time()
lib.hipsdp_comm_destroy(comm)
print("unique id %.2f s, comm init %.2f s, destroy %.2f s  [NCCL_SOCKET_IFNAME=%s NCCL_IB_DISABLE=%s]" %
      (t1 - t0, t2 - t1, time.time() - t2, os.environ.get("NCCL_SOCKET_IFNAME"), os.environ.get("NCCL_IB_DISABLE")))
