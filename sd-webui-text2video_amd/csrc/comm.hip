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

// ---- peer windows (round 6): device-initiated exchange over IPC-mapped mailboxes ------------------------------------------------
// An exchange of a T-sharded forward moves little data (512-byte statistics parts, one or two boundary frames, a reshard chunk) and
// there are 139 of them per forward: as RCCL group calls each costs a proxy round trip and a kernel of its own (15-35 us).  With a
// WINDOW attached to the communicator the same op is ONE small kernel of this library: every rank owns a window — hipMalloc'ed here,
// mapped into every peer with hipIpcOpenMemHandle (xGMI load / store path between the GPUs of a node) —
//     [flags: nranks x PX_NB words, 64 B apart][data: nranks sources x 2 slots x slot_bytes]
// and workgroup (j, q) of the exchange kernel
//   1. PUSHES its share j of this rank's message for peer q straight into q's window (system-scope write-through stores), fences, and
//      sets flag (me, j) in q's window to the pair's sequence number;
//   2. polls flag (q, j) in its OWN window (bounded: T2V_PEER_TIMEOUT_MS, default 20 s, then the fault word -> T2V_ERR_ASYNC and the
//      windows of this process switch off) and copies its share of q's message from the window to its destination.
// Pushes never wait for the peer, so there is no ordering between ranks to deadlock on.  Slot reuse: messages between a pair alternate
// between two slots by the pair's own sequence number k (counted on the host at enqueue time — both sides enqueue the same exchanges in
// the same order).  When r pushes message k + 1, r has finished exchange k, i.e. received q's message k, which q pushed (stream order)
// after its kernel of exchange k - 1 had completed: slot (k + 1) & 1 = (k - 1) & 1 is free.  Flags only grow (wait = "flag >= k").
// An op whose message does not fit a slot, or is not made of 16-byte units, takes the RCCL path: the two transports are independent.
constexpr int PX_NB = 8;            // workgroups (and flag words) per message
constexpr int PX_MAX_RANKS = 8;     // ranks of one communicator that can carry a window (a T group of 8 GPUs: 4)
constexpr size_t PX_FLAG_STRIDE = 64;

struct PxPeer {
  unsigned char* rslot;                    // peer's window: slot for messages from this rank
  unsigned long long* rflag;               // peer's window: flags (this rank, 0..PX_NB)
  const unsigned char* lslot;              // own window: slot with the peer's message
  const unsigned long long* lflag;         // own window: flags (peer, 0..PX_NB)
  unsigned long long seq;                  // the pair's sequence number of this message
  const unsigned char* ssrc[2];            // message to the peer = segment 0 | segment 1 (16-byte units, either may be empty)
  unsigned long long sbytes[2];
  unsigned char* rdst[2];                  // message from the peer, same form
  unsigned long long rbytes[2];
  int nb_send, nb_recv;                    // workgroups that share the message (a function of its size: both sides compute the same)
};
struct PxParams {
  PxPeer peer[PX_MAX_RANKS - 1];
  unsigned* fault;
  unsigned long long timeout_ticks;        // of the constant 100 MHz clock
};

struct PeerWindow {
  unsigned char* local = nullptr;
  unsigned char* peer[PX_MAX_RANKS] = {};
  size_t slot_bytes = 0, flag_bytes = 0, total = 0;
  unsigned long long pair_seq[PX_MAX_RANKS] = {};
  bool open = false;
  const char* mem_kind = "";             // uncached | finegrained | default (t2v_comm_impl_window_create)
  unsigned long long n_window_ops = 0, n_rccl_ops = 0;
};

struct t2v_comm {
  ncclComm_t comm;
  int nranks, rank;
  PeerWindow win;
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



// ---- the exchange kernel ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void px_st_sys(unsigned char* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 px_ld_sys(const unsigned char* p) {      // completion: px_wait4 below
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void px_wait4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
inline int px_nb_host(size_t bytes) {
  const size_t n = bytes >> 12;
  return n < 1 ? 1 : (n > (size_t)PX_NB ? PX_NB : (int)n);
}

__global__ __launch_bounds__(256) void peer_exchange_kernel(const PxParams p) {
  const PxPeer& P = p.peer[blockIdx.y];
  const int j = blockIdx.x, tid = threadIdx.x;
  // 1. push
  if (j < P.nb_send) {
    unsigned long long off = 0;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const unsigned long long units = P.sbytes[sg] >> 4;
      const unsigned long long u0 = units * j / P.nb_send, u1 = units * (j + 1) / P.nb_send;
      for (unsigned long long u = u0 + tid; u < u1; u += 256)
        px_st_sys(P.rslot + off + (u << 4), *reinterpret_cast<const f32x4*>(P.ssrc[sg] + (u << 4)));
      off += P.sbytes[sg];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(P.rflag + j * (PX_FLAG_STRIDE / 8), P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. wait for the peer's share j, copy it out of the window
  if (j < P.nb_recv) {
    __shared__ int ok;
    if (tid == 0) {
      const unsigned long long* f = P.lflag + j * (PX_FLAG_STRIDE / 8);
      unsigned polls = 0;
      unsigned long long t0 = 0;
      int good = 1;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < P.seq) {
        if (polls == 0) t0 = wall_clock64();
        if ((++polls & 63u) == 0u && wall_clock64() - t0 > p.timeout_ticks) {        // give up: flag it, never hang the device
          __hip_atomic_store(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          good = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      ok = good;
    }
    __syncthreads();
    if (!ok) return;
    unsigned long long off = 0;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const unsigned long long units = P.rbytes[sg] >> 4;
      const unsigned long long u0 = units * j / P.nb_recv, u1 = units * (j + 1) / P.nb_recv;
      for (unsigned long long u = u0 + tid; u < u1; u += 1024) {        // four system-scope loads in flight per lane
        f32x4 v[4] = {};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (u + 256 * k < u1) v[k] = px_ld_sys(P.lslot + off + ((u + 256 * k) << 4));
        px_wait4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (u + 256 * k < u1) *reinterpret_cast<f32x4*>(P.rdst[sg] + ((u + 256 * k) << 4)) = v[k];
      }
      off += P.rbytes[sg];
    }
  }
}

unsigned long long px_timeout_ticks() {
  static unsigned long long t = 0;
  if (t == 0) {
    const char* e = getenv("T2V_PEER_TIMEOUT_MS");
    const double ms = (e && *e) ? atof(e) : 20000.0;
    t = (unsigned long long)((ms > 1.0 ? ms : 1.0) * 1e5);      // 100 MHz
  }
  return t;
}

struct PxMsg {               // one peer's part of an exchange, as the host describes it
  int q;
  const void* ssrc[2]; size_t sbytes[2];
  void* rdst[2]; size_t rbytes[2];
};

// true: the op went over the window (launched on s; rc = its status).  false: not eligible (RCCL takes it) — nothing was launched, no
// state changed.  Eligibility must come out the same on every rank of the communicator, so it is decided from quantities that do not
// depend on the rank: `max_bytes` = the largest message ANY pair exchanges in this op, `unit` = a size every segment is a multiple of.
bool px_exchange(t2v_comm* c, const PxMsg* msgs, int n, size_t max_bytes, size_t unit, hipStream_t s, std::string& err, int& rc) {
  rc = T2V_OK;
  PeerWindow& w = c->win;
  if (!w.open || n <= 0 || n > PX_MAX_RANKS - 1 || max_bytes > w.slot_bytes || (unit & 15) != 0) return false;
  unsigned* fault = t2v_peer_fault_word();
  if (fault == nullptr || __atomic_load_n(fault, __ATOMIC_RELAXED) != 0u) return false;      // a peer wait timed out earlier: windows stay off
  for (int k = 0; k < n; ++k) {
    const PxMsg& m = msgs[k];
    bool bad = m.sbytes[0] + m.sbytes[1] > w.slot_bytes || m.rbytes[0] + m.rbytes[1] > w.slot_bytes;
    for (int sg = 0; sg < 2; ++sg)
      bad = bad || (m.sbytes[sg] & 15) || (m.rbytes[sg] & 15) || ((uintptr_t)m.ssrc[sg] & 15) || ((uintptr_t)m.rdst[sg] & 15);
    if (bad) { err = "peer exchange: a segment is not made of aligned 16-byte units or exceeds the slot (the op's own size bound is wrong)"; rc = T2V_ERR_BAD_ARG; return true; }
  }
  PxParams p;
  memset(&p, 0, sizeof p);
  p.fault = fault;
  p.timeout_ticks = px_timeout_ticks();
  int nbmax = 1;
  for (int k = 0; k < n; ++k) {
    const PxMsg& m = msgs[k];
    PxPeer& P = p.peer[k];
    const unsigned long long seq = ++w.pair_seq[m.q];
    const size_t slot = (size_t)(seq & 1);
    P.rslot = w.peer[m.q] + w.flag_bytes + ((size_t)c->rank * 2 + slot) * w.slot_bytes;
    P.rflag = reinterpret_cast<unsigned long long*>(w.peer[m.q] + (size_t)c->rank * PX_NB * PX_FLAG_STRIDE);
    P.lslot = w.local + w.flag_bytes + ((size_t)m.q * 2 + slot) * w.slot_bytes;
    P.lflag = reinterpret_cast<const unsigned long long*>(w.local + (size_t)m.q * PX_NB * PX_FLAG_STRIDE);
    P.seq = seq;
    for (int sg = 0; sg < 2; ++sg) {
      P.ssrc[sg] = static_cast<const unsigned char*>(m.ssrc[sg]);
      P.sbytes[sg] = m.sbytes[sg];
      P.rdst[sg] = static_cast<unsigned char*>(m.rdst[sg]);
      P.rbytes[sg] = m.rbytes[sg];
    }
    P.nb_send = px_nb_host(m.sbytes[0] + m.sbytes[1]);
    P.nb_recv = px_nb_host(m.rbytes[0] + m.rbytes[1]);
    nbmax = std::max(nbmax, std::max(P.nb_send, P.nb_recv));
  }
  hipLaunchKernelGGL(peer_exchange_kernel, dim3(nbmax, n), dim3(256), 0, s, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { err = std::string("peer exchange launch failed: ") + hipGetErrorString(e); rc = T2V_ERR_LAUNCH; }
  ++w.n_window_ops;
  return true;
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

// ---- window set-up ----------------------------------------------------------------------------------------------------------------
int t2v_comm_impl_window_create(t2v_comm* c, size_t slot_bytes, unsigned char handle_out[64], std::string& err) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (c->nranks > PX_MAX_RANKS) { err = "peer windows carry at most 8 ranks per communicator"; return T2V_ERR_BAD_ARG; }
  if (c->win.local) { err = "the communicator already has a window"; return T2V_ERR_BAD_ARG; }
  PeerWindow& w = c->win;
  w.slot_bytes = (slot_bytes + 255) & ~(size_t)255;
  w.flag_bytes = ((size_t)c->nranks * PX_NB * PX_FLAG_STRIDE + 4095) & ~(size_t)4095;
  w.total = w.flag_bytes + (size_t)c->nranks * 2 * w.slot_bytes;
  // Memory the PEERS write and this device polls: uncached device memory first (MTYPE UC — what RCCL itself allocates for its peer-written
  // buffers on gfx94x / gfx950: the L2 of this device must never hold a line a peer is about to overwrite through the fabric), then
  // fine-grained, then plain hipMalloc (coarse-grained; the kernel's system-scope sc0 sc1 accesses are then the only protection).
  // T2V_PEER_WINDOW_MEM=uncached | finegrained | default forces one.
  const char* forced = getenv("T2V_PEER_WINDOW_MEM");
  struct Kind { const char* name; int flags; };
  const Kind kinds[] = {{"uncached", (int)hipDeviceMallocUncached}, {"finegrained", (int)hipDeviceMallocFinegrained}, {"default", -1}};
  void* p = nullptr;
  hipIpcMemHandle_t h;
  const char* got = nullptr;
  for (const Kind& k : kinds) {
    if (forced && *forced && strcmp(forced, k.name) != 0) continue;
    void* q = nullptr;
    const hipError_t e = k.flags < 0 ? hipMalloc(&q, w.total) : hipExtMallocWithFlags(&q, w.total, (unsigned)k.flags);
    if (e != hipSuccess || q == nullptr) { (void)hipGetLastError(); continue; }
    if (hipMemset(q, 0, w.total) == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipIpcGetMemHandle(&h, q) == hipSuccess) {
      p = q;
      got = k.name;
      break;
    }
    (void)hipGetLastError();
    (void)hipFree(q);
  }
  if (p == nullptr) {
    err = "peer window: no allocation kind could be allocated and exported (hipIpcGetMemHandle; HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on hosts that "
          "only support dmabuf IPC)";
    return T2V_ERR_COMM;
  }
  w.mem_kind = got;
  w.local = static_cast<unsigned char*>(p);
  w.peer[c->rank] = w.local;
  memcpy(handle_out, &h, 64);
  return T2V_OK;
}

static void px_window_close(t2v_comm* c) {
  PeerWindow& w = c->win;
  if (!w.local) return;
  // nobody writes into this window any more: a peer's pushes of an exchange precede the flags this rank waited for in that exchange
  (void)hipDeviceSynchronize();
  for (int q = 0; q < c->nranks; ++q)
    if (q != c->rank && w.peer[q]) { (void)hipIpcCloseMemHandle(w.peer[q]); w.peer[q] = nullptr; }
  (void)hipFree(w.local);
  w.local = nullptr;
  w.open = false;
}

int t2v_comm_impl_window_open(t2v_comm* c, const unsigned char* handles, std::string& err) {
  PeerWindow& w = c->win;
  if (handles == nullptr) { px_window_close(c); return T2V_OK; }      // the group decided against windows (a rank could not map a peer)
  if (!w.local || w.open) { err = "window_open needs a window created on this communicator, once"; return T2V_ERR_BAD_ARG; }
  for (int q = 0; q < c->nranks; ++q) {
    if (q == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)q * 64, 64);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess || p == nullptr) {
      (void)hipGetLastError();
      for (int k = 0; k < q; ++k)
        if (k != c->rank && w.peer[k]) { (void)hipIpcCloseMemHandle(w.peer[k]); w.peer[k] = nullptr; }
      err = std::string("peer window: hipIpcOpenMemHandle failed for rank ") + std::to_string(q) + ": " + hipGetErrorString(e);
      return T2V_ERR_COMM;
    }
    w.peer[q] = static_cast<unsigned char*>(p);
  }
  w.open = true;
  return T2V_OK;
}

void t2v_comm_impl_counters(const t2v_comm* c, unsigned long long out[2]) {
  out[0] = c ? c->win.n_window_ops : 0;
  out[1] = c ? c->win.n_rccl_ops : 0;
}

const char* t2v_comm_impl_window_kind(const t2v_comm* c) { return (c && c->win.local) ? c->win.mem_kind : ""; }

void t2v_comm_impl_destroy(t2v_comm* c) {
  if (!c) return;
  px_window_close(c);
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
  {
    PxMsg m[PX_MAX_RANKS];
    int n = 0;
    for (int q = 0; q < nparts && nparts <= PX_MAX_RANKS; ++q)
      if (q != part) m[n++] = PxMsg{q, {b + (size_t)part * bytes, nullptr}, {bytes, 0}, {b + (size_t)q * bytes, nullptr}, {bytes, 0}};
    int prc;
    if (px_exchange(c, m, n, bytes, bytes, s, err, prc)) return prc;
    ++c->win.n_rccl_ops;
  }
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
  {
    PxMsg m[2];
    int n = 0;
    if (prev >= 0) m[n++] = PxMsg{prev, {b + frame_bytes, nullptr}, {frame_bytes, 0}, {b, nullptr}, {frame_bytes, 0}};
    if (next >= 0) m[n++] = PxMsg{next, {b + (size_t)F * frame_bytes, nullptr}, {frame_bytes, 0}, {b + (size_t)(F + 1) * frame_bytes, nullptr}, {frame_bytes, 0}};
    int prc;
    if (px_exchange(c, m, n, frame_bytes, frame_bytes, s, err, prc)) return prc;
    ++c->win.n_rccl_ops;
  }
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
  {
    PxMsg m[PX_MAX_RANKS];
    int n = 0;
    for (int q = 0; q < nparts && nparts <= PX_MAX_RANKS; ++q) {
      if (q == part) continue;
      PxMsg x{q, {pb + (size_t)part * part_bytes, nullptr}, {part_bytes, 0}, {pb + (size_t)q * part_bytes, nullptr}, {part_bytes, 0}};
      if (q == prev) { x.ssrc[1] = rb + frame_bytes; x.rdst[1] = rb; x.sbytes[1] = x.rbytes[1] = frame_bytes; }
      if (q == next) { x.ssrc[1] = rb + (size_t)F * frame_bytes; x.rdst[1] = rb + (size_t)(F + 1) * frame_bytes; x.sbytes[1] = x.rbytes[1] = frame_bytes; }
      m[n++] = x;
    }
    int prc;
    if (px_exchange(c, m, n, part_bytes + frame_bytes, (part_bytes | frame_bytes) & 15 ? 1 : 16, s, err, prc)) return prc;
    ++c->win.n_rccl_ops;
  }
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
  {
    PxMsg m[PX_MAX_RANKS];
    int n = 0;
    for (int q = 0; q < nparts && nparts <= PX_MAX_RANKS; ++q) {
      if (q == part) continue;
      if (dir == 0) m[n++] = PxMsg{q, {sb + (size_t)q * cnt(part) * chunk, nullptr}, {cnt(part) * chunk, 0}, {rb + (size_t)q * base_cnt * chunk, nullptr}, {cnt(q) * chunk, 0}};
      else m[n++] = PxMsg{q, {sb + (size_t)q * base_cnt * chunk, nullptr}, {cnt(q) * chunk, 0}, {rb + (size_t)q * cnt(part) * chunk, nullptr}, {cnt(part) * chunk, 0}};
    }
    int prc;
    if (px_exchange(c, m, n, (size_t)base_cnt * chunk, chunk, s, err, prc)) return prc;
    ++c->win.n_rccl_ops;
  }
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
