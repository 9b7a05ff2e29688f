// Multi-GPU form of the path, behind the C ABI: one process per GPU, the two exchanges as RCCL operations over xGMI on the
// handle's own stream, so that any host language - the C++ shim classes, the Python layer - shards k-modes and multipoles through the
// same three calls.
//   exchange 1  cpt_allgather_sources : rank r has integrated the modes k_all[r], k_all[r + W], ... (round-robin: every rank gets the
//               same mix of cheap low-k and expensive high-k modes); one ncclAllGather of the padded k-major blocks, then every rank
//               un-interleaves them into the full resident table [tp][nk_all][ntau].  The exchange is real: the transfer stage
//               splines the sources across ALL k (tm.cpp:604-639).
//   exchange 2  cpt_gather_transfer   : rank r has the multipoles l_all[r], l_all[r + W], ...; point-to-point sends to rank 0
//               (ncclSend / ncclRecv in one group: xGMI is point-to-point, this is its native pattern), un-interleaved there into
//               [tt][nl_all][nq].  Only results travel: the C_l integral over q needs every q of one l, which each rank already holds.
// RCCL is bound at run time (dlopen of librccl.so, CPT_RCCL_PATH overrides): the library must use the SAME HIP runtime as the rest
// of the process, and a host such as PyTorch ships its own copy next to its own libamdhip64.
#include <dlfcn.h>

#include "cpt_internal.h"

namespace {
// the few RCCL entry points used, with the types of <rccl/rccl.h> (ncclResult_t = int, ncclComm_t = opaque pointer)
struct NcclUniqueId { char internal[CPT_COMM_ID_BYTES]; };
enum { kNcclFloat64 = 8 };
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};
Rccl* rccl() {
  static Rccl R;
  static bool tried = false;
  if (tried) return R.so ? &R : nullptr;
  tried = true;
  // A copy the process has loaded already wins (PyTorch ships its own librccl next to its own HIP runtime: a second RCCL in the process
  // would talk to another runtime): RTLD_NOLOAD first, then an explicit CPT_RCCL_PATH, then the loader's search path.
  const char* path = getenv("CPT_RCCL_PATH");
  for (const char* n : {"librccl.so.1", "librccl.so"}) {
    R.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (R.so) break;
  }
  if (!R.so && path && *path) { R.so = dlopen(path, RTLD_NOW | RTLD_GLOBAL); if (!R.so) R.why = dlerror(); }
  if (!R.so)
    for (const char* n : {"librccl.so.1", "librccl.so"}) {
      R.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (R.so) break;
      R.why = dlerror();
    }
  if (!R.so) return nullptr;
  auto sym = [&](const char* name) { return dlsym(R.so, name); };
  R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
  R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
  R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
  R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
  R.Send = (decltype(R.Send))sym("ncclSend");
  R.Recv = (decltype(R.Recv))sym("ncclRecv");
  R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
  R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
  R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd) {
    R.why = "librccl.so lacks one of the entry points";
    dlclose(R.so);
    R.so = nullptr;
    return nullptr;
  }
  return &R;
}
#define CPT_RCCL(h, call)                                                                                     \
  do {                                                                                                        \
    const int r__ = (call);                                                                                   \
    if (r__ != 0) return cpt_fail(h, CPT_ERR_RUNTIME, "%s failed: %s", #call, R->GetErrorString ? R->GetErrorString(r__) : "RCCL error"); \
  } while (0)

// local k-major block [ntp][n_local][ntau] <-> padded block [ntp][n_max][ntau] (rows beyond n_local zero)
__global__ void k_pad_rows(const double* __restrict__ in, double* __restrict__ out, int ntp, int n_local, int n_max, int ntau) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)ntp * n_max * ntau;
  if (i >= total) return;
  const int it = (int)(i % ntau);
  const size_t r = i / ntau;
  const int j = (int)(r % n_max), tp = (int)(r / n_max);
  out[i] = (j < n_local) ? in[((size_t)tp * n_local + j) * ntau + it] : 0.;
}
// gathered blocks [world][ntp][n_max][ntau] -> full table [ntp][n_all][ntau] with row r + j * world <- block r, row j
__global__ void k_uninterleave(const double* __restrict__ blocks, double* __restrict__ full, int world, int ntp, int n_max, int n_all, int ntau) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)ntp * n_all * ntau;
  if (i >= total) return;
  const int it = (int)(i % ntau);
  const size_t r = i / ntau;
  const int row = (int)(r % n_all), tp = (int)(r / n_all);
  const int rank = row % world, j = row / world;
  full[i] = blocks[(((size_t)rank * ntp + tp) * n_max + j) * ntau + it];
}
}  // namespace

// shard r of a range of n items dealt round-robin over `world` ranks has (n - r + world - 1) / world items
static inline int shard_size(int n, int rank, int world) { return (n - rank + world - 1) / world; }

extern "C" {

int cpt_comm_get_unique_id(void* id) {
  if (!id) return CPT_ERR_INVALID;
  Rccl* R = rccl();
  if (!R) return cpt_fail(nullptr, CPT_ERR_NO_DEVICE, "RCCL is not available: %s", rccl() ? "" : "dlopen(librccl.so) failed (set CPT_RCCL_PATH)");
  NcclUniqueId u;
  const int rc = R->GetUniqueId(&u);
  if (rc != 0) return cpt_fail(nullptr, CPT_ERR_RUNTIME, "ncclGetUniqueId failed: %s", R->GetErrorString ? R->GetErrorString(rc) : "RCCL error");
  memcpy(id, &u, CPT_COMM_ID_BYTES);
  return CPT_OK;
}

int cpt_comm_init(cpt_handle* h, const void* id, int rank, int world) {
  if (!h) return CPT_ERR_INVALID;
  h->err.clear();
  if (!id || world < 1 || rank < 0 || rank >= world) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_comm_init (rank %d of %d)", rank, world);
  Rccl* R = rccl();
  if (!R) return cpt_fail(h, CPT_ERR_NO_DEVICE, "RCCL is not available: dlopen(librccl.so) failed (set CPT_RCCL_PATH)");
  if (h->comm) return cpt_fail(h, CPT_ERR_INVALID, "the handle already has a communicator");
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != h->device) (void)hipSetDevice(h->device);
  NcclUniqueId u;
  memcpy(&u, id, CPT_COMM_ID_BYTES);
  void* comm = nullptr;
  const int rc = R->CommInitRank(&comm, world, u, rank);
  if (prev != h->device && prev >= 0) (void)hipSetDevice(prev);
  if (rc != 0) return cpt_fail(h, CPT_ERR_RUNTIME, "ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(rc) : "RCCL error");
  h->comm = comm; h->comm_rank = rank; h->comm_world = world;
  return CPT_OK;
}

int cpt_comm_destroy(cpt_handle* h) {
  if (!h) return CPT_ERR_INVALID;
  if (h->comm) {
    Rccl* R = rccl();
    if (R) (void)R->CommDestroy(h->comm);
    h->comm = nullptr;
  }
  h->comm_world = 1; h->comm_rank = 0;
  return CPT_OK;
}

int cpt_allgather_sources(cpt_handle* h, int nk_all, int ntau) {
  if (!h) return CPT_ERR_INVALID;
  h->err.clear();
  const int world = h->comm_world, rank = h->comm_rank, ntp = h->cfg.tp_size;
  if (!h->comm) return cpt_fail(h, CPT_ERR_INVALID, "cpt_allgather_sources: no communicator (cpt_comm_init)");
  const int n_local = shard_size(nk_all, rank, world), n_max = shard_size(nk_all, 0, world);
  if (!h->d_src || h->src_nk != n_local || h->src_ntau != ntau)
    return cpt_fail(h, CPT_ERR_INVALID, "cpt_allgather_sources: the handle holds resident sources for %d k-modes x %d times, rank %d of %d needs %d x %d",
                    h->src_nk, h->src_ntau, rank, world, n_local, ntau);
  Rccl* R = rccl();
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != h->device) (void)hipSetDevice(h->device);
  struct Restore { int prev, dev; ~Restore() { if (prev != dev && prev >= 0) (void)hipSetDevice(prev); } } restore{prev, h->device};
  const size_t block = (size_t)ntp * n_max * ntau, full = (size_t)ntp * nk_all * ntau;
  int rc;
  if ((rc = cpt_reserve(h, &h->d_xsend, &h->xsend_cap, block))) return rc;
  if ((rc = cpt_reserve(h, &h->d_xrecv, &h->xrecv_cap, block * world))) return rc;
  hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((block + 255) / 256)), dim3(256), 0, h->stream, h->d_src, h->d_xsend, ntp, n_local, n_max, ntau);
  CPT_HIP(h, hipGetLastError());
  CPT_RCCL(h, R->AllGather(h->d_xsend, h->d_xrecv, block, kNcclFloat64, h->comm, h->stream));
  // (d_src is the input of k_pad_rows, which has been enqueued before: growing it now is safe only after the stream has drained)
  if (h->src_cap < full) {
    CPT_HIP(h, hipStreamSynchronize(h->stream));
    if ((rc = cpt_reserve(h, &h->d_src, &h->src_cap, full))) return rc;
  }
  hipLaunchKernelGGL(k_uninterleave, dim3((unsigned)((full + 255) / 256)), dim3(256), 0, h->stream, h->d_xrecv, h->d_src, world, ntp, n_max, nk_all, ntau);
  CPT_HIP(h, hipGetLastError());
  h->src_nk = nk_all; h->src_ntau = ntau;
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  return CPT_OK;
}

// rank r holds rows r, r + W, ... of a table [nb][n_all][ninner] as a block [nb][n_local][ninner]: point-to-point sends to rank 0 (ncclSend /
// ncclRecv in one group), un-interleaved there into the full table
static int gather_rows(cpt_handle* h, const char* who, const double* local_dev, int nb, int n_all, int ninner, double* full_dev) {
  const int world = h->comm_world, rank = h->comm_rank;
  if (!h->comm) return cpt_fail(h, CPT_ERR_INVALID, "%s: no communicator (cpt_comm_init)", who);
  if (!local_dev || (rank == 0 && !full_dev) || n_all < 1 || ninner < 1 || nb < 1) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to %s", who);
  Rccl* R = rccl();
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != h->device) (void)hipSetDevice(h->device);
  struct Restore { int prev, dev; ~Restore() { if (prev != dev && prev >= 0) (void)hipSetDevice(prev); } } restore{prev, h->device};
  const int n_local = shard_size(n_all, rank, world), n_max = shard_size(n_all, 0, world);
  const size_t block = (size_t)nb * n_max * ninner;
  int rc;
  if ((rc = cpt_reserve(h, &h->d_xsend, &h->xsend_cap, block))) return rc;
  if (rank == 0 && (rc = cpt_reserve(h, &h->d_xrecv, &h->xrecv_cap, block * world))) return rc;
  hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((block + 255) / 256)), dim3(256), 0, h->stream, local_dev, h->d_xsend, nb, n_local, n_max, ninner);
  CPT_HIP(h, hipGetLastError());
  // (a group that was started is always ended, whatever happens inside it: an open group would leave the peers hanging in theirs)
  int e = R->GroupStart(), e_end = 0;
  if (e == 0) {
    if (rank == 0) { for (int r = 1; r < world && e == 0; r++) e = R->Recv(h->d_xrecv + (size_t)r * block, block, kNcclFloat64, r, h->comm, h->stream); }
    else e = R->Send(h->d_xsend, block, kNcclFloat64, 0, h->comm, h->stream);
    e_end = R->GroupEnd();
  }
  if (e != 0 || e_end != 0) {
    (void)hipStreamSynchronize(h->stream);
    return cpt_fail(h, CPT_ERR_RUNTIME, "%s: RCCL send / receive failed: %s", who, R->GetErrorString ? R->GetErrorString(e ? e : e_end) : "RCCL error");
  }
  if (rank == 0) {
    CPT_HIP(h, hipMemcpyAsync(h->d_xrecv, h->d_xsend, block * sizeof(double), hipMemcpyDeviceToDevice, h->stream));   // own block
    const size_t full = (size_t)nb * n_all * ninner;
    hipLaunchKernelGGL(k_uninterleave, dim3((unsigned)((full + 255) / 256)), dim3(256), 0, h->stream, h->d_xrecv, full_dev, world, nb, n_max, n_all, ninner);
    CPT_HIP(h, hipGetLastError());
  }
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  return CPT_OK;
}

int cpt_gather_transfer(cpt_handle* h, const double* transfer_local_dev, int nl_all, int nq, double* transfer_full_dev) {
  if (!h) return CPT_ERR_INVALID;
  h->err.clear();
  return gather_rows(h, "cpt_gather_transfer", transfer_local_dev, h->cfg.tt_size, nl_all, nq, transfer_full_dev);
}

// the C_l rows of this rank's multipoles [nl_local][ct_size] -> rank 0: [nl_all][ct_size].  The C_l integral over q needs every q of one l, which the
// rank that projected that multipole holds: the spectra are finished where the transfer functions are, and 7 numbers per multipole travel
// instead of the transfer table (exchange 2)
int cpt_gather_cl(cpt_handle* h, const double* cl_local_dev, int nl_all, int ct_size, double* cl_full_dev) {
  if (!h) return CPT_ERR_INVALID;
  h->err.clear();
  return gather_rows(h, "cpt_gather_cl", cl_local_dev, 1, nl_all, ct_size, cl_full_dev);
}

// unit-test hooks (no communicator needed): the packing and un-interleaving kernels of the two exchanges on caller-provided device buffers
int cpt_dbg_pad_rows(cpt_handle* h, const double* in_dev, double* out_dev, int nbatch, int n_local, int n_max, int ninner) {
  if (!h || !in_dev || !out_dev) return CPT_ERR_INVALID;
  const size_t total = (size_t)nbatch * n_max * ninner;
  hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, in_dev, out_dev, nbatch, n_local, n_max, ninner);
  CPT_HIP(h, hipGetLastError());
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  return CPT_OK;
}
int cpt_dbg_uninterleave(cpt_handle* h, const double* blocks_dev, double* full_dev, int world, int nbatch, int n_max, int n_all, int ninner) {
  if (!h || !blocks_dev || !full_dev || world < 1) return CPT_ERR_INVALID;
  const size_t total = (size_t)nbatch * n_all * ninner;
  hipLaunchKernelGGL(k_uninterleave, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, blocks_dev, full_dev, world, nbatch, n_max, n_all, ninner);
  CPT_HIP(h, hipGetLastError());
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  return CPT_OK;
}

}  // extern "C"
