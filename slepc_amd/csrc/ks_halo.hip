// Halo exchange of MatMult through peer-mapped ghost mailboxes ("neighbour P2P of boundary x entries").
//
// The provider path (ks_spmv.hip) packs the boundary entries of x into a send buffer and hands it to the communicator - grouped
// ncclSend / ncclRecv, i.e. a library call and its kernels per product. Here the pack kernel writes each boundary entry STRAIGHT into
// the ghost mailbox of the rank that needs it (the mailbox is device memory of that rank, mapped here: hipIpc between processes,
// the plain pointer between ranks of one process, xGMI underneath between GPUs) and the receiver's unpack kernel copies its mailbox
// into the ghost array the off-diagonal rows read. No library call, no host involvement, everything on the halo stream under the
// diagonal-block product.
//
// Protocol, per matrix. Products are numbered seq = 1, 2, ...; parity = seq & 1 selects one of two data slots in every mailbox.
//   sender    pack:   wait until every destination has acknowledged product seq - 2 (its ack word in MY mailbox, written by that
//                     destination: the slot of parity seq & 1 is free again) -> gather x and store the values into the destinations'
//                     slots with system-scope stores -> every wave drains its stores, one lane per workgroup fences at system scope and
//                     takes a ticket -> the workgroup with the last ticket stamps flag[parity][me] = seq in every destination's mailbox.
//   receiver  unpack: wait until every source has stamped flag[parity] = seq -> copy the slot into the ghost array (ordinary device
//                     memory: the mailbox itself is uncached, so that a remote store is what the next load sees) -> the last workgroup
//                     writes ack = seq into every source's mailbox.
// The acknowledgement makes the two slots safe for ANY communication pattern and any call sequence (back-to-back products with no
// reduction in between, non-symmetric patterns): a sender is never more than two products ahead of a receiver. Every wait is bounded
// (KSGPU_ONESHOT_TIMEOUT_MS, default 2000): a rank that gives up poisons its ghosts with NaN and raises the context's error word,
// which the next host wait reports as KS_ERR_LIB - never a hang.
#include "ksgpu_internal.h"
#include <unistd.h>

namespace {

constexpr size_t HM_ALIGN = 256;
size_t hm_flag_off(int nghost) { return ((size_t)2 * (size_t)std::max(nghost, 1) * sizeof(double) + HM_ALIGN - 1) / HM_ALIGN * HM_ALIGN; }
size_t hm_ack_off(int nghost) { return hm_flag_off(nghost) + (size_t)2 * KS_HALO_MAX_PEERS * sizeof(unsigned long long); }
size_t hm_bytes(int nghost) { return hm_ack_off(nghost) + (size_t)KS_HALO_MAX_PEERS * sizeof(unsigned long long); }

struct HaloArgs {
  int npeers, nsend, nghost;
  unsigned long long seq;
  long long timeout_ticks;
  int *err, *err_local;
  int send_off[KS_HALO_MAX_PEERS + 1], send_cnt[KS_HALO_MAX_PEERS], recv_cnt[KS_HALO_MAX_PEERS];
  double *rdata[KS_HALO_MAX_PEERS];                    // where my segment goes in peer i's slot of this product's parity
  unsigned long long *rflag[KS_HALO_MAX_PEERS];        // my flag word of this parity in peer i's mailbox
  unsigned long long *rack[KS_HALO_MAX_PEERS];         // my ack word in peer i's mailbox
  const unsigned long long *lflag;                     // my mailbox: flag[parity][.]
  const unsigned long long *lack;                      // my mailbox: ack[.]
  const double *ldata;                                 // my mailbox: data slot of this parity
};

__device__ __forceinline__ bool wait_word(const unsigned long long *w, unsigned long long want, bool at_least, long long timeout_ticks)
{
  const long long t0 = wall_clock64();
  unsigned spins = 0;
  for (;;) {
    const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (at_least ? v >= want : v == want) return true;
    if ((++spins & 255u) == 0 && wall_clock64() - t0 > timeout_ticks) return false;
    __builtin_amdgcn_s_sleep(2);
  }
}

__global__ __launch_bounds__(256) void k_halo_pack(HaloArgs a, const int *__restrict__ send_idx, const double *__restrict__ x, unsigned *__restrict__ ticket)
{
  __shared__ int failed;
  __shared__ int last;
  const int tid = threadIdx.x;
  if (tid == 0) failed = *(volatile int *)a.err_local;              // an earlier product of this rank gave up: send (the peers may be fine), wait for nothing
  __syncthreads();
  if (tid < a.npeers && a.send_cnt[tid] > 0 && a.seq > 2 && !failed)
    if (!wait_word(a.lack + tid, a.seq - 2, true, a.timeout_ticks)) failed = 1;
  __syncthreads();
  for (int e = blockIdx.x * blockDim.x + tid; e < a.nsend; e += gridDim.x * blockDim.x) {
    int i = 0;
    while (i + 1 < a.npeers && e >= a.send_off[i + 1]) i++;
    __hip_atomic_store(a.rdata[i] + (e - a.send_off[i]), x[send_idx[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // every storing wave, before the workgroup's one fence
  __syncthreads();
  if (tid == 0) {
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    if (failed && *(volatile int *)a.err_local == 0) { *(volatile int *)a.err_local = 1; __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
  __syncthreads();
  if (!last) return;
  if (tid == 0) *ticket = 0;
  __threadfence();
  if (tid < a.npeers && a.send_cnt[tid] > 0) __hip_atomic_store(a.rflag[tid], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_halo_unpack(HaloArgs a, double *__restrict__ ghost, unsigned *__restrict__ ticket)
{
  __shared__ int failed;
  __shared__ int last;
  const int tid = threadIdx.x;
  if (tid == 0) failed = *(volatile int *)a.err_local;
  __syncthreads();
  if (tid < a.npeers && a.recv_cnt[tid] > 0 && !failed)
    if (!wait_word(a.lflag + tid, a.seq, false, a.timeout_ticks)) failed = 1;
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  for (int e = blockIdx.x * blockDim.x + tid; e < a.nghost; e += gridDim.x * blockDim.x)
    ghost[e] = failed ? nan : __hip_atomic_load(a.ldata + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    if (failed && *(volatile int *)a.err_local == 0) { *(volatile int *)a.err_local = 1; __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
  __syncthreads();
  if (!last) return;
  if (tid == 0) *ticket = 0;
  __threadfence();
  // the slot is free again: tell every source (also after a failure: the sources must not wait for this rank on top of it)
  if (tid < a.npeers && a.recv_cnt[tid] > 0) __hip_atomic_store(a.rack[tid], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct HaloHello {
  int ok, pid, device, npeers, nghost, nohandle;
  unsigned long long ptr;
  hipIpcMemHandle_t handle;
  int peers[KS_HALO_MAX_PEERS], recv_off[KS_HALO_MAX_PEERS], recv_cnt[KS_HALO_MAX_PEERS], send_cnt[KS_HALO_MAX_PEERS];
};

} // namespace

// Release in the order that keeps every remote store inside live memory: this rank's own streams are idle (the callers synchronise them), but a
// neighbour's last unpack may still be writing its acknowledgement into THIS rank's mailbox, and the neighbours hold this mailbox mapped.
//   collective = true (ks_mat_set_halo, every rank is in the call): barrier (all ranks' streams idle: nobody stores any more) -> close the imported
//     mailboxes -> barrier (nobody maps mine any more) -> free mine.
//   collective = false (ks_mat_destroy without a ks_mat_set_halo(A, KS_HALO_PROVIDER) before it): wait, bounded, until every destination has
//     acknowledged this rank's last product - the only store a neighbour can still owe this mailbox - then release. The neighbours' mappings of the
//     freed mailbox are theirs to close; the documented way to take the peer halo down is the collective call (include/ksgpu.h).
static void halo_close_imports(ks_mat A)
{
  auto &h = A->hp;
  for (int i = 0; i < KS_HALO_MAX_PEERS; i++) {
    if (h.opened[i] && h.peer_base[i]) hipIpcCloseMemHandle(h.peer_base[i]);
    h.peer_base[i] = nullptr; h.opened[i] = false;
  }
}
static void halo_free_own(ks_mat A)
{
  auto &h = A->hp;
  if (h.mine) hipFree(h.mine);
  h.mine = nullptr;
  if (h.tickets) hipFree(h.tickets);
  h.tickets = nullptr;
  (void)hipGetLastError();
}
static void halo_wait_last_acks(ks_mat A)
{
  auto &h = A->hp;
  if (!h.mine || h.seq == 0) return;
  const int np = std::min((int)A->peers.size(), KS_HALO_MAX_PEERS);
  unsigned long long acks[KS_HALO_MAX_PEERS];
  for (int spin = 0; spin < 2000; spin++) {                 // at most ~2 s, the peer waits' own limit
    if (hipMemcpy(acks, h.mine + hm_ack_off(A->nghost), sizeof(acks), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    bool all = true;
    for (int i = 0; i < np; i++) if (A->send_cnt[i] > 0 && acks[i] < h.seq) all = false;
    if (all) return;
    usleep(1000);
  }
}
void ks_halo_release(ks_mat A)
{
  auto &h = A->hp;
  if (h.enabled) halo_wait_last_acks(A);
  h.enabled = false;
  halo_close_imports(A);
  halo_free_own(A);
}

static int halo_err_words(ks_ctx ctx)
{
  auto &c = ctx->comm;
  if (c.halo_err_host) return KS_SUCCESS;
  KS_HIP(hipHostMalloc((void **)&c.halo_err_host, sizeof(int), hipHostMallocMapped));
  *c.halo_err_host = 0;
  KS_HIP(hipHostGetDevicePointer((void **)&c.halo_err_dev, c.halo_err_host, 0));
  KS_HIP(hipMalloc((void **)&c.halo_err_local, sizeof(int)));
  KS_HIP(hipMemset(c.halo_err_local, 0, sizeof(int)));
  return KS_SUCCESS;
}

// Collective over the ranks of the matrix's context. KS_HALO_PEER is taken only if EVERY rank could map the mailboxes of all its
// neighbours (agreed through an allgather); otherwise everything is released and the provider's exchange stays.
extern "C" int ks_mat_set_halo(ks_mat A, int kind, int *active)
{
  KS_CHECK(A, KS_ERR_ARG_NULL, "Mat is NULL");
  KS_CHECK(kind == KS_HALO_PROVIDER || kind == KS_HALO_PEER, KS_ERR_ARG_OUTOFRANGE, "unknown halo kind %d", kind);
  ks_ctx ctx = A->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  KS_HIP(ks_sync(ctx));
  if (ctx->halo_stream) KS_HIP(hipStreamSynchronize(ctx->halo_stream));
  if (active) *active = KS_HALO_PROVIDER;
  if (A->hp.enabled && ctx->comm.size > 1 && ctx->comm.ops.allgather_host) {
    // the peer halo is on, on every rank (it is only ever switched on by agreement): take it down in step with the others
    std::vector<int> tok(ctx->comm.size, 0); int one = 1;
    KS_CALL(ks_comm_allgather_host(ctx, &one, (int)sizeof(int), tok.data()));       // every rank's streams are idle: no store into any mailbox is pending
    A->hp.enabled = false;
    halo_close_imports(A);
    KS_CALL(ks_comm_allgather_host(ctx, &one, (int)sizeof(int), tok.data()));       // nobody maps this rank's mailbox any more
    halo_free_own(A);
  } else ks_halo_release(A);
  if (kind == KS_HALO_PROVIDER || ctx->comm.size <= 1 || A->shell_mult) return KS_SUCCESS;
  KS_CHECK(ctx->comm.ops.allgather_host, KS_ERR_ORDER, "no communicator installed");
  const int size = ctx->comm.size, rank = ctx->comm.rank, np = (int)A->peers.size();
  auto &h = A->hp;
  HaloHello me; memset(&me, 0, sizeof(me));
  me.pid = (int)getpid(); me.device = ctx->device; me.npeers = np; me.nghost = A->nghost;
  bool ok = np <= KS_HALO_MAX_PEERS && halo_err_words(ctx) == KS_SUCCESS;
  if (ok) {
    for (int i = 0; i < np; i++) { me.peers[i] = A->peers[i]; me.recv_off[i] = A->recv_off[i]; me.recv_cnt[i] = A->recv_cnt[i]; me.send_cnt[i] = A->send_cnt[i]; }
    h.bytes = hm_bytes(A->nghost);
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, h.bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    if (!p && hipExtMallocWithFlags(&p, h.bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    h.mine = (char *)p;
    ok = p && hipMemset(p, 0, h.bytes) == hipSuccess && hipMalloc((void **)&h.tickets, 2 * sizeof(unsigned)) == hipSuccess
         && hipMemset(h.tickets, 0, 2 * sizeof(unsigned)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (ok) {
      me.ptr = (unsigned long long)(uintptr_t)p;
      if (hipIpcGetMemHandle(&me.handle, p) != hipSuccess) { (void)hipGetLastError(); me.nohandle = 1; }
    }
  }
  me.ok = ok ? 1 : 0;
  std::vector<HaloHello> all(size);
  KS_CALL(ks_comm_allgather_host(ctx, &me, (int)sizeof(me), all.data()));
  for (int i = 0; i < np && ok; i++) {
    const HaloHello &o = all[A->peers[i]];
    if (!o.ok) { ok = false; break; }
    int j = -1;
    for (int q = 0; q < o.npeers; q++) if (o.peers[q] == rank) { j = q; break; }
    if (j < 0 || o.recv_cnt[j] != A->send_cnt[i] || o.send_cnt[j] != A->recv_cnt[i]) { ok = false; break; }     // the two halo plans must mirror each other
    h.ridx[i] = j; h.remote_off[i] = o.recv_off[j]; h.remote_nghost[i] = o.nghost;
    if (o.pid == me.pid) {
      if (o.device != ctx->device) {
        hipError_t e = hipDeviceEnablePeerAccess(o.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
        (void)hipGetLastError();
      }
      h.peer_base[i] = (char *)(uintptr_t)o.ptr;
    } else {
      void *q = nullptr;
      if (o.nohandle || hipIpcOpenMemHandle(&q, o.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = false; }
      else { h.peer_base[i] = (char *)q; h.opened[i] = true; }
    }
  }
  int mine_ok = ok ? 1 : 0;
  std::vector<int> oks(size, 0);
  KS_CALL(ks_comm_allgather_host(ctx, &mine_ok, (int)sizeof(int), oks.data()));
  for (int r = 0; r < size; r++) if (!oks[r]) ok = false;
  if (!ok) { ks_halo_release(A); return KS_SUCCESS; }
  int khz = 100000;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
  const char *tm = getenv("KSGPU_ONESHOT_TIMEOUT_MS");
  const long long ms = tm && atoll(tm) > 0 ? atoll(tm) : 2000;
  h.timeout_ticks = ms * khz;
  h.seq = 0;
  h.enabled = true;
  if (active) *active = KS_HALO_PEER;
  return KS_SUCCESS;
}

extern "C" int ks_mat_get_halo(ks_mat A, int *active)
{
  KS_CHECK(A && active, KS_ERR_ARG_NULL, "NULL argument");
  *active = A->hp.enabled ? KS_HALO_PEER : KS_HALO_PROVIDER;
  return KS_SUCCESS;
}

// the halo of one product on stream hs: pack into the neighbours' mailboxes, then wait for and unpack this rank's own
int ks_halo_peer_exchange(ks_mat A, const double *x, hipStream_t hs)
{
  ks_ctx ctx = A->ctx;
  auto &h = A->hp;
  const int np = (int)A->peers.size();
  HaloArgs a; memset(&a, 0, sizeof(a));
  a.npeers = np; a.nsend = A->nsend; a.nghost = A->nghost; a.seq = ++h.seq; a.timeout_ticks = h.timeout_ticks;
  a.err = ctx->comm.halo_err_dev; a.err_local = ctx->comm.halo_err_local;
  const int par = (int)(a.seq & 1ull);
  a.send_off[0] = 0;
  for (int i = 0; i < np; i++) {
    a.send_off[i] = A->send_off[i]; a.send_cnt[i] = A->send_cnt[i]; a.recv_cnt[i] = A->recv_cnt[i];
    char *pb = h.peer_base[i];
    const int rng = h.remote_nghost[i];
    a.rdata[i] = (double *)pb + (size_t)par * (size_t)std::max(rng, 1) + h.remote_off[i];
    a.rflag[i] = (unsigned long long *)(pb + hm_flag_off(rng)) + (size_t)par * KS_HALO_MAX_PEERS + h.ridx[i];
    a.rack[i] = (unsigned long long *)(pb + hm_ack_off(rng)) + h.ridx[i];
  }
  a.send_off[np] = A->nsend;
  a.ldata = (const double *)h.mine + (size_t)par * (size_t)std::max(A->nghost, 1);
  a.lflag = (const unsigned long long *)(h.mine + hm_flag_off(A->nghost)) + (size_t)par * KS_HALO_MAX_PEERS;
  a.lack = (const unsigned long long *)(h.mine + hm_ack_off(A->nghost));
  if (A->nsend > 0) {
    const int gb = std::max(1, std::min((A->nsend + 2047) / 2048, 32));
    hipLaunchKernelGGL(k_halo_pack, dim3(gb), dim3(256), 0, hs, a, A->send_idx, x, h.tickets);
  }
  if (A->nghost > 0) {
    const int gb = std::max(1, std::min((A->nghost + 2047) / 2048, 32));
    hipLaunchKernelGGL(k_halo_unpack, dim3(gb), dim3(256), 0, hs, a, A->ghost, h.tickets + 1);
  }
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
