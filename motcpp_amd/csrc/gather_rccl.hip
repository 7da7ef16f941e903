// Device-resident gather of the per-rank track tables over RCCL (xGMI), SURVEY.md §8(e): the data path of the trackers never
// crosses ranks — streams are independent, rank r owns streams [r*S, (r+1)*S) — so the only exchange is the final table
// gather. Each rank holds a PACKED table on the device (mot_bt_step_packed / mot_bt_device_output: rows [total][8], counts [S]);
// the gather is
//   1. ncclAllGather of the S per-stream counts (4*S bytes per rank) — every rank learns every rank's row total,
//   2. one grouped set of ncclBroadcast, root r sending exactly its rows (no padding: xGMI links are point to point,
//      bytes on a link are what the ring pays for),
// all on the context's stream: no D2H -> H2D bounce, and tracker kernels of the next frame queue behind it.
// RCCL is bound at run time (dlopen) so that the kernel library carries no link-time dependency on a 570 MB collective library
// and shares the copy the host process already loaded (PyTorch ships its own librccl.so).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/motcpp_amd.h"
#include "ctx.hpp"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)  // a copy the process already mapped (PyTorch's) wins: one collective runtime per process
      if (!x.lib) x.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (!x.lib) x.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!x.lib) x.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!x.lib) { x.why = std::string("librccl.so.1 not found: ") + dlerror(); return x; }
#define MOT_SYM(f) x.f = reinterpret_cast<decltype(x.f)>(dlsym(x.lib, "nccl" #f)); if (!x.f) { x.why = "symbol nccl" #f " missing"; x.lib = nullptr; return x; }
    MOT_SYM(GetUniqueId) MOT_SYM(CommInitRank) MOT_SYM(CommDestroy) MOT_SYM(AllGather) MOT_SYM(Broadcast) MOT_SYM(GroupStart)
    MOT_SYM(GroupEnd) MOT_SYM(GetErrorString)
#undef MOT_SYM
    return x;
  }();
  return &r;
}

}  // namespace

struct mot_comm {
  mot_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  int* d_counts_all = nullptr;  // [world][S]
  int counts_cap = 0;
  std::vector<int> h_counts;
};

#define MOT_NCCL(c, expr)                                                                   \
  do {                                                                                      \
    const ncclResult_t r_ = (expr);                                                         \
    if (r_ != ncclSuccess) { (c)->err = std::string(#expr) + ": " + rccl()->GetErrorString(r_); return MOT_ERR_HIP; } \
  } while (0)
#define MOT_GH(c, expr)                                                                     \
  do {                                                                                      \
    const hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) { (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return MOT_ERR_HIP; } \
  } while (0)

extern "C" {

int mot_comm_unique_id(void* id128) {
  Rccl* R = rccl();
  if (!R->lib || !id128) return MOT_ERR_INVALID;
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return MOT_ERR_HIP;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return MOT_OK;
}

int mot_comm_create(mot_ctx* ctx, int world, int rank, const void* id128, mot_comm** out) {
  if (!ctx || !out || !id128 || world < 1 || rank < 0 || rank >= world) return MOT_ERR_INVALID;
  Rccl* R = rccl();
  if (!R->lib) { ctx->err = "mot_comm_create: " + R->why; return MOT_ERR_INVALID; }
  MOT_GH(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  auto* c = new mot_comm;
  c->ctx = ctx; c->world = world; c->rank = rank;
  const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + R->GetErrorString(r); delete c; return MOT_ERR_HIP; }
  *out = c;
  return MOT_OK;
}

int mot_comm_destroy(mot_comm* c) {
  if (!c) return MOT_OK;
  if (c->d_counts_all) (void)hipFree(c->d_counts_all);
  if (c->comm) (void)rccl()->CommDestroy(c->comm);
  delete c;
  return MOT_OK;
}

int mot_comm_gather_tables(mot_comm* c, const float* d_rows, const int* d_counts, int nstreams, float* d_rows_all, int rows_cap,
                           int* h_counts_all, int* h_rank_rows) {
  if (!c || !d_counts || nstreams <= 0 || !d_rows_all || !h_counts_all) return MOT_ERR_INVALID;
  mot_ctx* ctx = c->ctx;
  Rccl* R = rccl();
  hipStream_t st = ctx->stream;
  const int W = c->world, S = nstreams;
  if (W * S > c->counts_cap) {
    if (c->d_counts_all) (void)hipFree(c->d_counts_all);
    c->d_counts_all = nullptr; c->counts_cap = 0;
    MOT_GH(ctx, hipMalloc(reinterpret_cast<void**>(&c->d_counts_all), sizeof(int) * static_cast<size_t>(W) * S));
    c->counts_cap = W * S;
  }
  // 1. every rank's per-stream counts
  MOT_NCCL(ctx, R->AllGather(d_counts, c->d_counts_all, static_cast<size_t>(S), ncclInt32, c->comm, st));
  MOT_GH(ctx, hipMemcpyAsync(h_counts_all, c->d_counts_all, sizeof(int) * static_cast<size_t>(W) * S, hipMemcpyDeviceToHost, st));
  MOT_GH(ctx, hipStreamSynchronize(st));
  std::vector<long long> off(static_cast<size_t>(W) + 1, 0);
  for (int r = 0; r < W; ++r) {
    long long t = 0;
    for (int s = 0; s < S; ++s) t += h_counts_all[static_cast<size_t>(r) * S + s];
    off[r + 1] = off[r] + t;
    if (h_rank_rows) h_rank_rows[r] = static_cast<int>(t);
  }
  if (off[W] > rows_cap) { ctx->err = "mot_comm_gather_tables: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  // 2. exact-size exchange: rank r's rows land at row offset off[r] on every rank
  MOT_NCCL(ctx, R->GroupStart());
  for (int r = 0; r < W; ++r) {
    const size_t cnt = static_cast<size_t>(off[r + 1] - off[r]) * 8;
    if (cnt == 0) continue;
    float* dst = d_rows_all + static_cast<size_t>(off[r]) * 8;  // (non-root ranks pass the destination as the unused send buffer)
    const ncclResult_t e = R->Broadcast(r == c->rank ? d_rows : dst, dst, cnt, ncclFloat32, r, c->comm, st);
    if (e != ncclSuccess) { (void)R->GroupEnd(); ctx->err = std::string("ncclBroadcast: ") + R->GetErrorString(e); return MOT_ERR_HIP; }
  }
  MOT_NCCL(ctx, R->GroupEnd());
  return MOT_OK;
}

}  // extern "C"
