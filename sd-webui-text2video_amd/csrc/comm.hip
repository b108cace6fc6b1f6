// Communicators for the T-axis (frame) sharding of a UNet forward: one process per GPU, RCCL over xGMI.
//
// The reference's only collective is an all-gather of decoded samples (lvdm/utils/dist_utils.py:13-19); the exchanges
// here are the ones a frame-sharded forward needs before its temporal ops: all-gather of GroupNorm
// statistics partials and of temporal-attention K/V, and a +-1 frame neighbour exchange for the (3,1,1) convolutions.
// They are ops of the denoise program and run on the launch stream between the kernels, so a sharded forward stays one
// host call.  RCCL is dlopen'ed on first use: the library has no link-time dependency on it (single-GPU use never loads
// it, and the process may already hold torch's copy — the same soname resolves to that one).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "t2v_kernels.h"

struct t2v_comm {
  ncclComm_t comm;
  int nranks, rank;
};

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl g_rccl;
std::once_flag g_once;

template <typename F>
bool sym(void* h, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(h, name));
  return fn != nullptr;
}

void load_rccl() {
  // T2V_RCCL_SONAME: use exactly this library (deployments with a private RCCL build; tests force the failure path with it)
  const char* forced = getenv("T2V_RCCL_SONAME");
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  std::string last;
  for (const char* n : names) {
    if (forced && *forced) n = forced;
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
    const char* e = dlerror();      // dlerror() clears the message: read it exactly once per failure
    last = std::string(n) + ": " + (e ? e : "?");
    if (forced && *forced) break;
  }
  if (!g_rccl.handle) {
    g_rccl.err = "cannot dlopen librccl (" + last + ")";
    return;
  }
  void* h = g_rccl.handle;
  const bool ok = sym(h, "ncclGetUniqueId", g_rccl.GetUniqueId) && sym(h, "ncclCommInitRank", g_rccl.CommInitRank) &&
                  sym(h, "ncclCommDestroy", g_rccl.CommDestroy) && sym(h, "ncclAllGather", g_rccl.AllGather) &&
                  sym(h, "ncclSend", g_rccl.Send) && sym(h, "ncclRecv", g_rccl.Recv) &&
                  sym(h, "ncclGroupStart", g_rccl.GroupStart) && sym(h, "ncclGroupEnd", g_rccl.GroupEnd) &&
                  sym(h, "ncclGetErrorString", g_rccl.GetErrorString);
  if (!ok) {
    g_rccl.err = "librccl lacks an expected nccl* symbol";
    g_rccl.handle = nullptr;
  }
}

const Rccl* rccl() {
  std::call_once(g_once, load_rccl);
  return g_rccl.handle ? &g_rccl : nullptr;
}

}  // namespace

const char* t2v_comm_load_error() { return g_rccl.err.c_str(); }

int t2v_comm_impl_unique_id(unsigned char id[128], std::string& err) {
  const Rccl* r = rccl();
  if (!r) { err = g_rccl.err; return T2V_ERR_COMM; }
  ncclUniqueId u;
  const ncclResult_t rc = r->GetUniqueId(&u);
  if (rc != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + r->GetErrorString(rc); return T2V_ERR_COMM; }
  static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id, &u, 128);
  return T2V_OK;
}

int t2v_comm_impl_create(const unsigned char id[128], int nranks, int rank, t2v_comm** out, std::string& err) {
  const Rccl* r = rccl();
  if (!r) { err = g_rccl.err; return T2V_ERR_COMM; }
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclComm_t c;
  const ncclResult_t rc = r->CommInitRank(&c, nranks, u, rank);
  if (rc != ncclSuccess) { err = std::string("ncclCommInitRank: ") + r->GetErrorString(rc); return T2V_ERR_COMM; }
  *out = new t2v_comm{c, nranks, rank};
  return T2V_OK;
}

void t2v_comm_impl_destroy(t2v_comm* c) {
  if (!c) return;
  const Rccl* r = rccl();
  if (r) (void)r->CommDestroy(c->comm);
  delete c;
}

int t2v_comm_impl_size(const t2v_comm* c) { return c ? c->nranks : 0; }

// In-place all-gather: part q at base + q*bytes, this rank's part already written by the preceding kernels.
int t2v_comm_allgather(t2v_comm* c, void* base, size_t bytes, int nparts, int part, hipStream_t s, std::string& err) {
  if (nparts <= 1 && !c) return T2V_OK;     // single slice without a communicator: nothing to do
  const Rccl* r = rccl();
  if (!r || !c) { err = c ? g_rccl.err : "collective op in a plan without a communicator (t2v_plan_set_comm)"; return T2V_ERR_COMM; }
  if (c->nranks != nparts || c->rank != part) { err = "all-gather parts do not match the communicator (nranks / rank)"; return T2V_ERR_BAD_ARG; }
  unsigned char* b = static_cast<unsigned char*>(base);
  const ncclResult_t rc = r->AllGather(b + (size_t)part * bytes, b, bytes, ncclUint8, c->comm, s);
  if (rc != ncclSuccess) { err = std::string("ncclAllGather: ") + r->GetErrorString(rc); return T2V_ERR_COMM; }
  return T2V_OK;
}

int t2v_comm_impl_all_gather(t2v_comm* c, void* base, size_t bytes, hipStream_t s, std::string& err) {
  return t2v_comm_allgather(c, base, bytes, c->nranks, c->rank, s, err);
}

// Token buffer of F+2 frames: send frame 1 to prev / frame F to next, receive into frame 0 / frame F+1.
int t2v_comm_halo(t2v_comm* c, void* base, size_t frame_bytes, int F, int prev, int next, hipStream_t s, std::string& err) {
  if (prev < 0 && next < 0) return T2V_OK;
  const Rccl* r = rccl();
  if (!r || !c) { err = c ? g_rccl.err : "collective op in a plan without a communicator (t2v_plan_set_comm)"; return T2V_ERR_COMM; }
  if (prev >= c->nranks || next >= c->nranks || prev == c->rank || next == c->rank) { err = "halo exchange: bad neighbour rank"; return T2V_ERR_BAD_ARG; }
  unsigned char* b = static_cast<unsigned char*>(base);
  ncclResult_t rc = r->GroupStart();
  if (rc == ncclSuccess && prev >= 0) rc = r->Send(b + frame_bytes, frame_bytes, ncclUint8, prev, c->comm, s);
  if (rc == ncclSuccess && prev >= 0) rc = r->Recv(b, frame_bytes, ncclUint8, prev, c->comm, s);
  if (rc == ncclSuccess && next >= 0) rc = r->Send(b + (size_t)F * frame_bytes, frame_bytes, ncclUint8, next, c->comm, s);
  if (rc == ncclSuccess && next >= 0) rc = r->Recv(b + (size_t)(F + 1) * frame_bytes, frame_bytes, ncclUint8, next, c->comm, s);
  const ncclResult_t rc2 = r->GroupEnd();
  if (rc == ncclSuccess) rc = rc2;
  if (rc != ncclSuccess) { err = std::string("halo exchange (ncclSend/ncclRecv): ") + r->GetErrorString(rc); return T2V_ERR_COMM; }
  return T2V_OK;
}

// Statistics parts of a cross-frame GroupNorm to every rank AND the raw boundary frames of the temporal convolution behind it to the
// two neighbours, as ONE group of point-to-point transfers (T2V_OP_STATS_HALO).  parts: [nparts][part_bytes], this rank's part
// in place; raw: halo-padded buffer of F + 2 frames whose interior the preceding kernels wrote.  Per peer the order of the
// transfers is the same on both sides (part first, then the frame), which is how RCCL matches several sends to one peer.
int t2v_comm_stats_halo(t2v_comm* c, void* parts, size_t part_bytes, int nparts, int part, void* raw, size_t frame_bytes, int F,
                        int prev, int next, hipStream_t s, std::string& err) {
  if (nparts <= 1 && !c) return T2V_OK;
  const Rccl* r = rccl();
  if (!r || !c) { err = c ? g_rccl.err : "collective op in a plan without a communicator (t2v_plan_set_comm)"; return T2V_ERR_COMM; }
  if (c->nranks != nparts || c->rank != part) { err = "statistics + halo exchange: parts do not match the communicator (nranks / rank)"; return T2V_ERR_BAD_ARG; }
  if (prev >= nparts || next >= nparts || prev == part || next == part) { err = "statistics + halo exchange: bad neighbour rank"; return T2V_ERR_BAD_ARG; }
  unsigned char* pb = static_cast<unsigned char*>(parts);
  unsigned char* rb = static_cast<unsigned char*>(raw);
  ncclResult_t rc = r->GroupStart();
  for (int q = 0; q < nparts && rc == ncclSuccess; ++q) {
    if (q == part) continue;
    rc = r->Send(pb + (size_t)part * part_bytes, part_bytes, ncclUint8, q, c->comm, s);
    if (rc == ncclSuccess) rc = r->Recv(pb + (size_t)q * part_bytes, part_bytes, ncclUint8, q, c->comm, s);
  }
  if (rc == ncclSuccess && prev >= 0) rc = r->Send(rb + frame_bytes, frame_bytes, ncclUint8, prev, c->comm, s);
  if (rc == ncclSuccess && prev >= 0) rc = r->Recv(rb, frame_bytes, ncclUint8, prev, c->comm, s);
  if (rc == ncclSuccess && next >= 0) rc = r->Send(rb + (size_t)F * frame_bytes, frame_bytes, ncclUint8, next, c->comm, s);
  if (rc == ncclSuccess && next >= 0) rc = r->Recv(rb + (size_t)(F + 1) * frame_bytes, frame_bytes, ncclUint8, next, c->comm, s);
  const ncclResult_t rc2 = r->GroupEnd();
  if (rc == ncclSuccess) rc = rc2;
  if (rc != ncclSuccess) { err = std::string("statistics + halo exchange (ncclSend/ncclRecv): ") + r->GetErrorString(rc); return T2V_ERR_COMM; }
  return T2V_OK;
}

// Frame <-> pixel resharding of a T-sharded clip (slice q holds cnt(q) frames: base_cnt, a shorter last one).  chunk =
// bytes of one frame's share for one rank (hw / R rows).  dir 0 (frames -> pixels): this rank sends cnt(me) chunks to every
// peer q from send + q * cnt(me) * chunk (packed by T2V_OP_RESHARD_ROWS) and receives cnt(q) chunks from q straight into
// recv + q * base_cnt * chunk (frames of q are consecutive in the pixel-sharded layout).  dir 1 (pixels -> frames): the
// mirror image — sends cnt(q) chunks to q from send + q * base_cnt * chunk, receives cnt(me) chunks from q into
// recv + q * cnt(me) * chunk (unpacked afterwards).  The rank's own part is moved by the reshard ops, not here.
int t2v_comm_alltoall(t2v_comm* c, void* send, void* recv, size_t chunk, int nparts, int part, int base_cnt, int last_cnt, int dir,
                      hipStream_t s, std::string& err) {
  if (nparts <= 1 && !c) return T2V_OK;
  const Rccl* r = rccl();
  if (!r || !c) { err = c ? g_rccl.err : "collective op in a plan without a communicator (t2v_plan_set_comm)"; return T2V_ERR_COMM; }
  if (c->nranks != nparts || c->rank != part) { err = "all-to-all parts do not match the communicator (nranks / rank)"; return T2V_ERR_BAD_ARG; }
  auto cnt = [&](int q) { return (size_t)(q == nparts - 1 ? last_cnt : base_cnt); };
  unsigned char* sb = static_cast<unsigned char*>(send);
  unsigned char* rb = static_cast<unsigned char*>(recv);
  ncclResult_t rc = r->GroupStart();
  for (int q = 0; q < nparts && rc == ncclSuccess; ++q) {
    if (q == part) continue;
    if (dir == 0) {
      rc = r->Send(sb + (size_t)q * cnt(part) * chunk, cnt(part) * chunk, ncclUint8, q, c->comm, s);
      if (rc == ncclSuccess) rc = r->Recv(rb + (size_t)q * base_cnt * chunk, cnt(q) * chunk, ncclUint8, q, c->comm, s);
    } else {
      rc = r->Send(sb + (size_t)q * base_cnt * chunk, cnt(q) * chunk, ncclUint8, q, c->comm, s);
      if (rc == ncclSuccess) rc = r->Recv(rb + (size_t)q * cnt(part) * chunk, cnt(part) * chunk, ncclUint8, q, c->comm, s);
    }
  }
  const ncclResult_t rc2 = r->GroupEnd();
  if (rc == ncclSuccess) rc = rc2;
  if (rc != ncclSuccess) { err = std::string("all-to-all (ncclSend/ncclRecv): ") + r->GetErrorString(rc); return T2V_ERR_COMM; }
  return T2V_OK;
}
