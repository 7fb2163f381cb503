// vlb_comm_*: the data-parallel gradient exchange of a C / C++ host, RCCL over xGMI.  Replaces the all-reduce that
// torch DistributedDataParallel / apex DDP issue for the reference (pretrain/function/train.py:89-90,353-354; one SUM all-reduce of the
// gradients per optimizer step, §8e) for a caller that is not a torch program.  The Python host of this repo issues the same
// collectives through torch.distributed (parallel.py: one process group, one launcher); both bind the same RCCL entry points on the same
// flat-buffer slices, and nothing in the kernels depends on who issues them.
//
// RCCL is bound at RUN time (dlopen / dlsym), never at link time: inside a torch process the copy torch already loaded is reused
// (RTLD_NOLOAD first), so one process never holds two RCCL instances; a process without RCCL can still load this library for
// everything else.  VLB_RCCL_PATH names another librccl.so.
//
// One communicator = one rank = one GPU (hipSetDevice before vlb_comm_init).  All collectives are enqueued on the caller's stream, in
// place where RCCL allows it, and never synchronise; `dtype`: 0 = fp32, 1 = the library's 16-bit type (bf16; fp16 in the f16 build).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "vlb_common.h"

namespace {

struct UniqueId { char internal[128]; };       // == ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value to ncclCommInitRank
enum { kSum = 0, kFloat16 = 6, kFloat32 = 7, kBfloat16 = 9 };     // ncclRedOp_t / ncclDataType_t values of rccl.h

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;

template <typename F>
bool sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  if (!f) vlb_set_error("vlb_comm: %s not found in the RCCL library", name);
  return f != nullptr;
}

bool load_rccl() {
  if (g_rccl.handle) return true;
  const char* env = getenv("VLB_RCCL_PATH");
  void* h = nullptr;
  if (env && *env) {
    h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  } else {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy this process already holds (torch's)
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) {
    vlb_set_error("vlb_comm: cannot load RCCL (%s); set VLB_RCCL_PATH", dlerror());
    return false;
  }
  g_rccl.handle = h;
  const bool ok = sym(g_rccl.GetUniqueId, "ncclGetUniqueId") && sym(g_rccl.CommInitRank, "ncclCommInitRank") &&
                  sym(g_rccl.CommDestroy, "ncclCommDestroy") && sym(g_rccl.AllReduce, "ncclAllReduce") &&
                  sym(g_rccl.ReduceScatter, "ncclReduceScatter") && sym(g_rccl.AllGather, "ncclAllGather") &&
                  sym(g_rccl.GetErrorString, "ncclGetErrorString");
  if (!ok) g_rccl.handle = nullptr;
  return ok;
}

int check(int rc, const char* what) {
  if (rc == 0) return 0;
  vlb_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return VLB_ERR_HIP;
}

struct Comm {
  void* nccl;
  int rank, world;
};

int nccl_dtype(int dtype) { return dtype == 0 ? kFloat32 : (VLB_ACT_IS_F16 ? kFloat16 : kBfloat16); }

}  // namespace

// 128 opaque bytes that rank 0 creates and hands to every rank through the host's own channel (file, socket, MPI, torch store)
extern "C" int vlb_comm_unique_id(void* id128) {
  VLB_CHECK_ARG(id128 != nullptr, "vlb_comm_unique_id: null id buffer");
  if (!load_rccl()) return VLB_ERR_HIP;
  UniqueId id;
  if (int e = check(g_rccl.GetUniqueId(&id), "vlb_comm_unique_id")) return e;
  memcpy(id128, &id, sizeof(id));
  return 0;
}

// collective over all `world` ranks (each on its own current device); *comm receives the handle for the calls below
extern "C" int vlb_comm_init(int rank, int world, const void* id128, void** comm) {
  VLB_CHECK_ARG(comm != nullptr && id128 != nullptr, "vlb_comm_init: null argument");
  VLB_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "vlb_comm_init: rank %d outside world %d", rank, world);
  if (!load_rccl()) return VLB_ERR_HIP;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  void* c = nullptr;
  if (int e = check(g_rccl.CommInitRank(&c, world, id, rank), "vlb_comm_init")) return e;
  *comm = new Comm{c, rank, world};
  return 0;
}

// in-place SUM over the ranks of buf[count] -- one gradient bucket (contiguous slice of the flat gradient / of its 16-bit wire image)
extern "C" int vlb_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, hipStream_t stream) {
  VLB_CHECK_ARG(comm != nullptr, "vlb_comm_allreduce_bucket: null communicator");
  VLB_CHECK_ARG(count >= 0 && (dtype == 0 || dtype == 1), "vlb_comm_allreduce_bucket: bad count / dtype");
  if (count == 0) return 0;
  VLB_CHECK_ARG(buf != nullptr, "vlb_comm_allreduce_bucket: null buffer");
  Comm* c = static_cast<Comm*>(comm);
  return check(g_rccl.AllReduce(buf, buf, (size_t)count, nccl_dtype(dtype), kSum, c->nccl, stream), "vlb_comm_allreduce_bucket");
}

// sharded optimizer, first half: out[count / world] <- this rank's slice of the SUM of buf[count]  (count % world == 0)
extern "C" int vlb_comm_reduce_scatter_bucket(void* comm, const void* buf, void* out, long count, int dtype, hipStream_t stream) {
  VLB_CHECK_ARG(comm != nullptr, "vlb_comm_reduce_scatter_bucket: null communicator");
  Comm* c = static_cast<Comm*>(comm);
  VLB_CHECK_ARG(count >= 0 && count % c->world == 0 && (dtype == 0 || dtype == 1), "vlb_comm_reduce_scatter_bucket: count %ld must be a multiple of the world size %d", count, c->world);
  if (count == 0) return 0;
  VLB_CHECK_ARG(buf != nullptr && out != nullptr, "vlb_comm_reduce_scatter_bucket: null buffer");
  return check(g_rccl.ReduceScatter(buf, out, (size_t)(count / c->world), nccl_dtype(dtype), kSum, c->nccl, stream), "vlb_comm_reduce_scatter_bucket");
}

// sharded optimizer, second half: out[count] <- the ranks' slices in[count / world] in rank order (the updated 16-bit weights)
extern "C" int vlb_comm_allgather_bucket(void* comm, const void* in, void* out, long count, int dtype, hipStream_t stream) {
  VLB_CHECK_ARG(comm != nullptr, "vlb_comm_allgather_bucket: null communicator");
  Comm* c = static_cast<Comm*>(comm);
  VLB_CHECK_ARG(count >= 0 && count % c->world == 0 && (dtype == 0 || dtype == 1), "vlb_comm_allgather_bucket: count %ld must be a multiple of the world size %d", count, c->world);
  if (count == 0) return 0;
  VLB_CHECK_ARG(in != nullptr && out != nullptr, "vlb_comm_allgather_bucket: null buffer");
  return check(g_rccl.AllGather(in, out, (size_t)(count / c->world), nccl_dtype(dtype), c->nccl, stream), "vlb_comm_allgather_bucket");
}

extern "C" int vlb_comm_finalize(void* comm) {
  if (!comm) return 0;
  Comm* c = static_cast<Comm*>(comm);
  const int e = check(g_rccl.CommDestroy(c->nccl), "vlb_comm_finalize");
  delete c;
  return e;
}
