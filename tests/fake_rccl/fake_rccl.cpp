// TEST INFRASTRUCTURE — a stand-in for librccl that moves the bytes of the library's collective calls between PROCESSES SHARING
// ONE GPU (the build environment only has 1-GPU boxes, and RCCL itself refuses several ranks on one device).
//
// libt2v_hip.so resolves RCCL with dlopen; T2V_RCCL_SONAME points it at this library instead.  Every entry point the product
// uses is implemented with the semantics RCCL documents — ncclAllGather (in-place allowed), grouped ncclSend / ncclRecv (several transfers to one peer are matched in issue order) — on
// top of a POSIX shared-memory segment: data is staged device -> host -> shared memory -> host -> device, ordered with the
// caller's stream by hipStreamSynchronize and between ranks by a sense-reversing barrier in the segment.  What this proves is
// everything in csrc/comm.hip that is NOT RCCL itself: which bytes go to which peer at which offsets, in-place parts, uneven
// slices, call order across ranks (a mismatch deadlocks or corrupts and the test fails); what it cannot prove is RCCL's own
// behaviour and timing on xGMI.  Built by tests/test_gpu_fake_rccl.py with hipcc; never loaded by the product on its own.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct FakeComm;
typedef FakeComm* ncclComm_t;
}

namespace {
constexpr size_t MAX_RANKS = 8;
constexpr size_t SLOT = 32u << 20;          // bytes one rank may publish per call (all its sends / its all-gather part)

constexpr int MAX_MSGS = 4;                 // transfers one rank may queue for ONE peer inside one group (matched in issue order, as RCCL does)

struct Shared {
  std::atomic<int> arrived, generation, attached;
  int send_cnt[MAX_RANKS][MAX_RANKS];                                                            // [src][dst]
  size_t send_off[MAX_RANKS][MAX_RANKS][MAX_MSGS], send_len[MAX_RANKS][MAX_RANKS][MAX_MSGS];     // inside src's slot
};

struct PendingOp { bool send; void* ptr; size_t bytes; int peer; };
}  // namespace

struct FakeComm {
  int nranks, rank;
  Shared* sh;
  unsigned char* slots;                       // nranks x SLOT
  size_t map_bytes;
  std::vector<PendingOp> pending;
  hipStream_t stream = nullptr;
};

namespace {
thread_local int g_group_depth = 0;
thread_local FakeComm* g_group_comm = nullptr;

void barrier(FakeComm* c) {
  Shared* s = c->sh;
  const int gen = s->generation.load();
  if (s->arrived.fetch_add(1) == c->nranks - 1) {
    s->arrived.store(0);
    s->generation.fetch_add(1);
  } else {
    while (s->generation.load() == gen) usleep(50);
  }
}

ncclResult_t flush_group(FakeComm* c) {
  if (c == nullptr || c->pending.empty()) return ncclSuccess;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return ncclSystemError;
  // publish every send of this rank into its own slot
  size_t off = 0;
  for (int d = 0; d < c->nranks; ++d) c->sh->send_cnt[c->rank][d] = 0;
  for (const PendingOp& op : c->pending) {
    if (!op.send) continue;
    int& k = c->sh->send_cnt[c->rank][op.peer];
    if (off + op.bytes > SLOT || k >= MAX_MSGS) return ncclInvalidArgument;
    if (hipMemcpy(c->slots + (size_t)c->rank * SLOT + off, op.ptr, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
    c->sh->send_off[c->rank][op.peer][k] = off;
    c->sh->send_len[c->rank][op.peer][k] = op.bytes;
    ++k;
    off += op.bytes;
  }
  barrier(c);
  ncclResult_t rc = ncclSuccess;
  int taken[MAX_RANKS] = {0};                   // the k-th receive from a peer takes that peer's k-th send to this rank
  for (const PendingOp& op : c->pending) {
    if (op.send) continue;
    const int k = taken[op.peer]++;
    if (k >= c->sh->send_cnt[op.peer][c->rank] || c->sh->send_len[op.peer][c->rank][k] != op.bytes) { rc = ncclInvalidArgument; continue; }   // count / size mismatch between the two sides
    if (hipMemcpy(op.ptr, c->slots + (size_t)op.peer * SLOT + c->sh->send_off[op.peer][c->rank][k], op.bytes, hipMemcpyHostToDevice) != hipSuccess)
      rc = ncclSystemError;
  }
  for (int q = 0; q < c->nranks; ++q)
    if (q != c->rank && taken[q] != c->sh->send_cnt[q][c->rank]) rc = ncclInvalidArgument;       // a send nobody received
  barrier(c);                                   // nobody overwrites its slot before every peer has read it
  c->pending.clear();
  return rc;
}
}  // namespace

extern "C" {

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : (r == ncclInvalidArgument ? "invalid argument (fake rccl)" : "system error (fake rccl)"); }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/t2v_fake_rccl_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > (int)MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const size_t bytes = sizeof(Shared) + 4096 + (size_t)nranks * SLOT;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;
  } else {
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {      // wait for rank 0 to create and size it
      fd = shm_open(id.internal, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)) { close(fd); fd = -1; }
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) return ncclSystemError;
  }
  void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return ncclSystemError;
  FakeComm* c = new FakeComm();
  c->nranks = nranks; c->rank = rank; c->sh = reinterpret_cast<Shared*>(m);
  c->slots = reinterpret_cast<unsigned char*>(m) + ((sizeof(Shared) + 4095) / 4096) * 4096;
  c->map_bytes = bytes;
  if (c->sh->attached.fetch_add(1) == nranks - 1) shm_unlink(id.internal);    // everyone is attached: the name can go
  else while (c->sh->attached.load() < nranks) usleep(200);
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (c) { munmap(c->sh, c->map_bytes); delete c; }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (--g_group_depth > 0) return ncclSuccess;
  FakeComm* c = g_group_comm;
  g_group_comm = nullptr;
  return flush_group(c);
}

static ncclResult_t p2p(bool send, void* ptr, size_t count, int peer, ncclComm_t c, hipStream_t s) {
  if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
  c->stream = s;
  c->pending.push_back({send, ptr, count, peer});
  if (g_group_depth == 0) return flush_group(c);
  g_group_comm = c;
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return p2p(true, const_cast<void*>(buf), count, peer, c, s); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return p2p(false, buf, count, peer, c, s); }

ncclResult_t ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
  if (!c || count > SLOT) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclSystemError;
  if (hipMemcpy(c->slots + (size_t)c->rank * SLOT, sendbuf, count, hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
  barrier(c);
  ncclResult_t rc = ncclSuccess;
  for (int q = 0; q < c->nranks; ++q)
    if (hipMemcpy(static_cast<unsigned char*>(recvbuf) + (size_t)q * count, c->slots + (size_t)q * SLOT, count, hipMemcpyHostToDevice) != hipSuccess) rc = ncclSystemError;
  barrier(c);
  return rc;
}

}  // extern "C"
