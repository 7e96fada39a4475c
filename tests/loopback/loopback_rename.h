/* TEST INFRASTRUCTURE, never part of libbp5.so.  Force-included (-include) when bp5_device.hip is compiled for
 * libbp5_loopback.so: the nine RCCL entry points the library calls are renamed to the host-shared-memory transport of
 * loopback_rccl.cpp, so that N ranks of the library can run as N processes on ONE GPU (RCCL itself refuses two ranks
 * on one device).  Everything else -- pack/unpack kernels, exchange schedule, dot-product corrections, the all-reduce
 * placement -- is the product source, unchanged. */
#pragma once
#define ncclGetUniqueId bp5lb_ncclGetUniqueId
#define ncclCommInitRank bp5lb_ncclCommInitRank
#define ncclCommDestroy bp5lb_ncclCommDestroy
#define ncclGetErrorString bp5lb_ncclGetErrorString
#define ncclGroupStart bp5lb_ncclGroupStart
#define ncclGroupEnd bp5lb_ncclGroupEnd
#define ncclSend bp5lb_ncclSend
#define ncclRecv bp5lb_ncclRecv
#define ncclAllReduce bp5lb_ncclAllReduce
