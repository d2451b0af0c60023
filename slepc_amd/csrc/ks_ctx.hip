// Context, error reporting, HIP-event profiling and the allreduce providers (RCCL / callback).
#include "ksgpu_internal.h"
#include "ks_oneshot.cuh"
#include <cstdarg>
#include <atomic>
#include <chrono>
#include <dlfcn.h>
#include <unistd.h>
#include <cstdint>

static thread_local char g_errmsg[512] = "";

void ks_set_error(const char *fmt, ...)
{
  va_list ap; va_start(ap, fmt); vsnprintf(g_errmsg, sizeof(g_errmsg), fmt, ap); va_end(ap);
}

extern "C" const char *ks_last_error_message(void) { return g_errmsg; }

extern "C" const char *ks_error_string(int rc)
{
  switch (rc) {
    case KS_SUCCESS: return "success";
    case KS_ERR_MEM: return "out of memory";
    case KS_ERR_SUP: return "no support for the requested operation";
    case KS_ERR_ORDER: return "operation done in wrong order";
    case KS_ERR_ARG_SIZ: return "nonconforming object sizes";
    case KS_ERR_ARG_WRONG: return "wrong argument";
    case KS_ERR_ARG_OUTOFRANGE: return "argument out of range";
    case KS_ERR_USER_INPUT: return "invalid user input (invalid inner product)";
    case KS_ERR_MAT_LU_ZRPVT: return "zero pivot in LU factorization";
    case KS_ERR_ARG_WRONGSTATE: return "object in wrong state";
    case KS_ERR_ARG_INCOMP: return "arguments are incompatible";
    case KS_ERR_LIB: return "error in external library (HIP/RCCL)";
    case KS_ERR_PLIB: return "internal library error";
    case KS_ERR_CONV_FAILED: return "convergence failed";
    case KS_ERR_ARG_NULL: return "null argument";
    case KS_ERR_FILE_OPEN: return "unable to open file";
    case KS_ERR_FILE_UNEXPECTED: return "unexpected data in file";
    case KS_ERR_ARG_IDN: return "two arguments must be different";
    case KS_ERR_NOT_CONVERGED: return "linear solve did not converge";
    case KS_ERR_GPU: return "no usable gfx950 GPU";
    default: return "unknown error";
  }
}

// ---- which HIP runtime(s) this process maps ------------------------------------------------------------------------------------------
// PyTorch wheels bundle a libamdhip64.so whose SONAME is libamdhip64.so.7, the name this library NEEDs: loaded first, it is the runtime this
// library binds to; loaded after this library (whose RUNPATH finds /opt/rocm's copy), it is a SECOND runtime in the process, because torch
// asks for the file name libamdhip64.so, which matches no loaded SONAME.
static int mapped_hip_runtimes(char paths[][256], int max)
{
  FILE *f = fopen("/proc/self/maps", "r");
  if (!f) return 0;
  char line[1024]; int n = 0;
  while (fgets(line, sizeof(line), f)) {
    char *p = strchr(line, '/');
    if (!p) continue;
    char *nl = strchr(p, '\n'); if (nl) *nl = 0;
    const char *base = strrchr(p, '/'); base = base ? base + 1 : p;
    if (strncmp(base, "libamdhip64.so", 14) != 0) continue;
    bool seen = false;
    for (int i = 0; i < n; i++) if (!strcmp(paths[i], p)) { seen = true; break; }
    if (!seen && n < max) { snprintf(paths[n], 256, "%s", p); n++; }
  }
  fclose(f);
  return n;
}
static std::atomic<int> g_allow_two_runtimes{0};
extern "C" int ks_runtime_allow_multiple(int allow) { g_allow_two_runtimes.store(allow ? 1 : 0); return KS_SUCCESS; }
static std::atomic<long long> g_occ_failures{0};
static char g_occ_last[128] = "";
int ks_occupancy(const void *kernel, int block_threads, size_t lds_bytes, int fallback)
{
  int nb = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block_threads, lds_bytes);
  if (e == hipSuccess && nb >= 1) return nb;
  (void)hipGetLastError();                                   // a failed query must not show up as the next launch's error
  snprintf(g_occ_last, sizeof(g_occ_last), "%s (blocks %d, %d threads, %zu B LDS)", e == hipSuccess ? "success with 0 blocks" : hipGetErrorName(e), nb, block_threads, lds_bytes);
  if (g_occ_failures.fetch_add(1) == 0) {
    char paths[4][256]; const int np = mapped_hip_runtimes(paths, 4);
    fprintf(stderr, "libksgpu: warning: an occupancy query failed: %s; grids fall back to %d workgroup(s) per CU for that kernel. HIP runtimes mapped: %d", g_occ_last, fallback, np);
    for (int i = 0; i < np; i++) fprintf(stderr, "%s%s", i ? ", " : " (", paths[i]);
    fprintf(stderr, "%s\n", np ? ")" : "");
  }
  return fallback;
}
extern "C" int ks_runtime_info(char *json, int len)
{
  KS_CHECK(json && len > 0, KS_ERR_ARG_NULL, "NULL buffer");
  Dl_info di; memset(&di, 0, sizeof(di));
  const char *bound = (dladdr((void *)&hipGetDeviceCount, &di) && di.dli_fname) ? di.dli_fname : "?";
  int ver = 0; if (hipRuntimeGetVersion(&ver) != hipSuccess) { (void)hipGetLastError(); ver = 0; }
  char paths[4][256]; const int np = mapped_hip_runtimes(paths, 4);
  int o = snprintf(json, len, "{\"hip_runtime_path\": \"%s\", \"hip_runtime_version\": %d, \"hip_runtimes_mapped\": [", bound, ver);
  for (int i = 0; i < np && o < len; i++) o += snprintf(json + o, len - o, "%s\"%s\"", i ? ", " : "", paths[i]);
  if (o < len) o += snprintf(json + o, len - o, "], \"occupancy_query_failures\": %lld, \"occupancy_last_error\": \"%s\"}", (long long)g_occ_failures.load(), g_occ_last);
  KS_CHECK(o < len, KS_ERR_ARG_SIZ, "buffer of %d bytes too short", len);
  return KS_SUCCESS;
}

extern "C" int ks_ctx_create(int device, void *stream, ks_ctx *out)
{
  KS_CHECK(out, KS_ERR_ARG_NULL, "ctx output pointer is NULL");
  {
    char paths[4][256];
    if (mapped_hip_runtimes(paths, 4) > 1 && !g_allow_two_runtimes.load())
      KS_FAIL(KS_ERR_LIB, "two HIP runtimes are mapped in this process (%s and %s): each would drive the GPU through its own HSA runtime. Load the other HIP user "
              "(e.g. `import torch`) BEFORE libksgpu.so, which then binds to that runtime (same SONAME), or keep it out of this process", paths[0], paths[1]);
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) KS_FAIL(KS_ERR_GPU, "no HIP device available (%s); libksgpu has no CPU fallback", hipGetErrorString(e));
  KS_CHECK(device >= 0 && device < ndev, KS_ERR_ARG_OUTOFRANGE, "device %d out of range (have %d)", device, ndev);
  KS_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  KS_HIP(hipGetDeviceProperties(&prop, device));
  ks_ctx ctx = new ks_ctx_s();
  ctx->device = device;
  ctx->num_cu = prop.multiProcessorCount;
  snprintf(ctx->arch, sizeof(ctx->arch), "%s", prop.gcnArchName);
  ctx->mem_total = prop.totalGlobalMem;
  if (strncmp(ctx->arch, "gfx950", 6) != 0) {
    delete ctx;
    KS_FAIL(KS_ERR_GPU, "device %d is %s; libksgpu carries gfx950 code objects only", device, prop.gcnArchName);
  }
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; KS_FAIL(KS_ERR_LIB, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e)); }
    ctx->own_stream = true;
  }
  ctx->h_pinned_len = KS_PINNED_D2H_BYTES / sizeof(double) + 2 * KS_PINNED_H2D_DOUBLES;
  e = hipHostMalloc((void **)&ctx->h_pinned, ctx->h_pinned_len * sizeof(double), hipHostMallocMapped);
  if (e != hipSuccess) {
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    KS_FAIL(KS_ERR_MEM, "hipHostMalloc of the staging area failed: %s", hipGetErrorString(e));
  }
  memset(ctx->h_pinned, 0, ctx->h_pinned_len * sizeof(double));      // the results' stamp starts at 0: the first batch carries 1
  if (hipHostGetDevicePointer(&ctx->h_pinned_dev, ctx->h_pinned, 0) != hipSuccess) { (void)hipGetLastError(); ctx->h_pinned_dev = nullptr; }     // nullptr: results come back through the copy engine
  *out = ctx;
  return KS_SUCCESS;
}

extern "C" int ks_ctx_destroy(ks_ctx ctx)
{
  if (!ctx) return KS_SUCCESS;
  hipSetDevice(ctx->device);
  ks_sync(ctx);
  for (auto &p : ctx->pending) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
  for (auto &e : ctx->event_pool) hipEventDestroy(e);
  ks_oneshot_release(ctx);
  if (ctx->comm.halo_err_host) hipHostFree(ctx->comm.halo_err_host);
  if (ctx->comm.halo_err_local) hipFree(ctx->comm.halo_err_local);
  if (ctx->comm.nccl_comm && ctx->comm.rccl_lib) {
    typedef int (*destroy_t)(void *);
    destroy_t f = (destroy_t)dlsym(ctx->comm.rccl_lib, "ncclCommDestroy");
    if (f) f(ctx->comm.nccl_comm);
  }
  if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
  for (int i = 0; i < 2; i++) if (ctx->ev_h2d[i]) hipEventDestroy(ctx->ev_h2d[i]);
  if (ctx->ev_fetch) hipEventDestroy(ctx->ev_fetch);
  if (ctx->ev_mail) hipEventDestroy(ctx->ev_mail);
  if (ctx->gs_mail) hipHostFree(ctx->gs_mail);
  if (ctx->comm.ag_dev) hipFree(ctx->comm.ag_dev);
  if (ctx->split.dev) hipFree(ctx->split.dev);
  if (ctx->halo_stream) { hipStreamSynchronize(ctx->halo_stream); hipStreamDestroy(ctx->halo_stream); }
  if (ctx->ev_x) hipEventDestroy(ctx->ev_x);
  if (ctx->ev_halo) hipEventDestroy(ctx->ev_halo);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return KS_SUCCESS;
}

int ks_ctx_halo_stream(ks_ctx ctx)
{
  if (ctx->halo_stream) return KS_SUCCESS;
  KS_HIP(hipStreamCreateWithFlags(&ctx->halo_stream, hipStreamNonBlocking));
  KS_HIP(hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming));
  KS_HIP(hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
  return KS_SUCCESS;
}

extern "C" int ks_ctx_synchronize(ks_ctx ctx)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  const hipError_t e = ks_sync(ctx);
  if (e != hipSuccess) KS_CALL(ks_oneshot_error(ctx));        // a one-shot allreduce or a peer-mapped halo exchange that gave up: say which
  KS_HIP(e);
  return KS_SUCCESS;
}

#ifdef KSD_TEST_HOOKS
extern "C" int ks_ctx_set_debug(ks_ctx ctx, int key, long long value)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  switch (key) {
    case KS_DEBUG_NO_FUSED_GS: ctx->dbg.no_fused_gs = value != 0; break;
    case KS_DEBUG_NO_MFMA: ctx->dbg.no_mfma = value != 0; break;
    case KS_DEBUG_NO_SPMV_DOT: ctx->dbg.no_spmv_dot = value != 0; break;
    case KS_DEBUG_FORCE_MULTI: ctx->dbg.force_multi = value != 0; break;
    case KS_DEBUG_HALO_OVERLAP: ctx->halo_overlap = value != 0; break;
    case KS_DEBUG_ONESHOT_SEQ0: ctx->dbg.oneshot_seq0 = (unsigned)value; break;
    default: KS_FAIL(KS_ERR_ARG_OUTOFRANGE, "unknown debug key %d", key);
  }
  return KS_SUCCESS;
}
#endif

extern "C" int ks_ctx_sync_count(ks_ctx ctx, long long *count)
{
  KS_CHECK(ctx && count, KS_ERR_ARG_NULL, "NULL argument");
  *count = ctx->nsync;
  return KS_SUCCESS;
}

extern "C" int ks_ctx_device_info(ks_ctx ctx, char *arch, int arch_len, int *num_cu, size_t *mem_total)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", ctx->arch);
  if (num_cu) *num_cu = ctx->num_cu;
  if (mem_total) *mem_total = ctx->mem_total;
  return KS_SUCCESS;
}

// ---- profiling ----------------------------------------------------------------------------------
static const char *g_class_names[KS_K_COUNT] = {
  "spmv_csr", "bv_dot_sweep", "gs_bookkeeping", "gs_update_fused_dot", "gs_update", "bv_scale", "bv_multinplace", "bv_copy",
  "bv_mult", "bv_dot_panel", "bv_norm", "halo_exchange", "allreduce", "gated_noop", "other", "spmv_dot_fused" };

extern "C" const char *ks_prof_class_name(int k) { return (k >= 0 && k < KS_K_COUNT) ? g_class_names[k] : "?"; }

// The reference's log event (PetscLogEventRegister names, bvfunc.c:69-86; PETSc's own for the product's halo and the reduction) under which a
// -log_view of the reference shows the work of each class. A fused launch does the work of two events and says so.
static const char *g_event_names[KS_K_COUNT] = {
  "BVMatMultVec", "BVDotVec", "BVOrthogonalizeV", "BVMultVec+BVDotVec", "BVMultVec+BVScale", "BVScale", "BVMultInPlace", "BVCopy",
  "BVMult", "BVDot", "BVNormVec", "VecScatterBegin+VecScatterEnd", "MPIU_Allreduce (in BVDotVec)", "-", "-", "BVMatMultVec+BVDotVec" };
extern "C" const char *ks_prof_event_name(int k) { return (k >= 0 && k < KS_K_COUNT) ? g_event_names[k] : "?"; }

static int get_event(ks_ctx ctx, hipEvent_t *e)
{
  if (!ctx->event_pool.empty()) { *e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return KS_SUCCESS; }
  // timing only: no system-scope fence (cache write-back and invalidation) when the event completes - it would cost the launches around it
  // more than the event itself (hip_runtime_api.h, hipEventDisableSystemFence)
  if (hipEventCreateWithFlags(e, hipEventDisableSystemFence) != hipSuccess) { (void)hipGetLastError(); KS_HIP(hipEventCreate(e)); }
  return KS_SUCCESS;
}

int ks_prof_begin(ks_ctx ctx, int kclass, int variant, double bytes, double hbm)
{
  KsProfPending p; p.kclass = kclass; p.variant = (variant >= 0 && variant < KS_PROF_VARIANTS) ? variant : 0; p.bytes = bytes; p.hbm = hbm;
  p.tag_col = -1; p.tag_slot = 0; p.tag_k = 0; p.tag_n = 0; p.done = false;
  if (ctx->pending.size() > 400000) { ctx->prof_on = false; return KS_ERR_MEM; }   // runaway instrumentation: stop recording, keep running
  KS_CALL(get_event(ctx, &p.e0)); KS_CALL(get_event(ctx, &p.e1));
  KS_HIP(hipEventRecord(p.e0, ctx->stream));
  ctx->pending.push_back(p);
  return KS_SUCCESS;
}

int ks_prof_end(ks_ctx ctx, size_t index)
{
  if (index >= ctx->pending.size()) return KS_SUCCESS;
  ctx->pending[index].done = true;
  KS_HIP(hipEventRecord(ctx->pending[index].e1, ctx->stream));
  return KS_SUCCESS;
}

int ks_prof_flush(ks_ctx ctx)
{
  if (ctx->pending.empty()) return KS_SUCCESS;
  KS_HIP(ks_sync(ctx));
  for (auto &p : ctx->pending) {
    float ms = 0.f;
    if (!p.done) { (void)hipGetLastError(); }
    else if (hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) { (void)hipGetLastError(); }     // never leave a sticky error behind for the host application
    else {
      KsProfSlot &sl = ctx->prof[p.kclass][p.variant];
      sl.launches++; sl.ms += ms; sl.bytes += p.bytes; sl.hbm += p.hbm;
    }
    ctx->event_pool.push_back(p.e0); ctx->event_pool.push_back(p.e1);
  }
  ctx->pending.clear();
  return KS_SUCCESS;
}

extern "C" int ks_prof_enable(ks_ctx ctx, int on)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  if (!on) KS_CALL(ks_prof_flush(ctx));
  ctx->prof_on = on != 0;
  ctx->prof_mask = (on == 0 || on == 1) ? 0xffffffffu : ((unsigned)on >> 1);   // on>1: bit (class+1) selects a class
  return KS_SUCCESS;
}

extern "C" int ks_prof_reset(ks_ctx ctx)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  KS_CALL(ks_prof_flush(ctx));
  for (int i = 0; i < KS_K_COUNT; i++) for (int v = 0; v < KS_PROF_VARIANTS; v++) ctx->prof[i][v] = KsProfSlot();
  return KS_SUCCESS;
}

extern "C" int ks_prof_get(ks_ctx ctx, int kclass, int variant, long long *launches, double *ms, double *bytes, double *hbm)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  KS_CHECK(kclass >= 0 && kclass < KS_K_COUNT, KS_ERR_ARG_OUTOFRANGE, "kernel class %d out of range", kclass);
  KS_CHECK(variant < KS_PROF_VARIANTS, KS_ERR_ARG_OUTOFRANGE, "variant %d out of range", variant);
  KS_CALL(ks_prof_flush(ctx));
  KsProfSlot t;
  for (int v = 0; v < KS_PROF_VARIANTS; v++) {
    if (variant >= 0 && v != variant) continue;
    const KsProfSlot &sl = ctx->prof[kclass][v];
    t.launches += sl.launches; t.ms += sl.ms; t.bytes += sl.bytes; t.hbm += sl.hbm;
  }
  if (launches) *launches = t.launches;
  if (ms) *ms = t.ms;
  if (bytes) *bytes = t.bytes;
  if (hbm) *hbm = t.hbm;
  return KS_SUCCESS;
}

// Speculative Gram-Schmidt slots: the host learns from the step records which launches really ran.
//   update slot p of column c ran iff p <= passes(c); it was the fused (update + next-pass dots) form iff
//   p < passes(c) or the column needed the explicit-norm fallback; bookkeeping slot p ran iff p <= passes(c)
//   (+1 with the fallback). Everything else exited at its gate and is re-filed under KS_K_NOOP.
// Bytes: compulsory HBM traffic of an update = 8n(k+2) (read k columns + read/write v); the SURVEY 8d
// algorithmic figure of the fused form adds the gemv-C of the next pass it replaces: 8n(k+2) + 8n(k+1).
void ks_prof_resolve_gs(ks_ctx ctx, const KsStepRec *recs, int col0, int col1)
{
  for (auto &p : ctx->pending) {
    if (p.tag_col < col0 || p.tag_col > col1) continue;
    const KsStepRec &r = recs[p.tag_col - col0];
    const int passes = r.passes, expl = r.expl;
    const double n = (double)p.tag_n, k = (double)p.tag_k;
    if (p.kclass == KS_K_UPD_FUSED || p.kclass == KS_K_UPD) {
      if (p.tag_slot > passes) { p.kclass = KS_K_NOOP; p.bytes = 0.0; p.hbm = 0.0; }
      else if (p.tag_slot < passes || expl) { p.kclass = KS_K_UPD_FUSED; p.hbm = 8.0 * n * (k + (expl ? 2 : 1)); p.bytes = 8.0 * n * (k + 2) + 8.0 * n * (k + 1); }   // a fused pass that is not followed by an explicit norm keeps its result in registers: no write
      else { p.kclass = KS_K_UPD; p.hbm = 8.0 * n * (k + 2); p.bytes = p.hbm; }
    } else if (p.kclass == KS_K_GSFIN) {
      if (p.tag_slot > passes + (expl ? 1 : 0)) { p.kclass = KS_K_NOOP; p.bytes = 0.0; p.hbm = 0.0; }
    } else if (p.kclass == KS_K_SCALE) {
      if (!expl) { p.kclass = KS_K_NOOP; p.bytes = 0.0; p.hbm = 0.0; } else { p.bytes = p.hbm = 16.0 * n; }
    }
    p.tag_col = -1;
  }
}

// ---- communicator -------------------------------------------------------------------------------
// RCCL is resolved at run time: a launcher that already loaded librccl (torch.distributed's "nccl"
// backend IS RCCL) shares that copy; otherwise /opt/rocm/lib/librccl.so.1 is opened.
struct NcclUniqueIdRaw { char internal[KS_UNIQUE_ID_BYTES]; };
typedef int (*nccl_getuid_t)(NcclUniqueIdRaw *);
typedef int (*nccl_initrank_t)(void **, int, NcclUniqueIdRaw, int);
typedef int (*nccl_allreduce_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_allgather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_send_t)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_recv_t)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_group_t)();
typedef int (*nccl_bcast_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
enum { NCCL_INT8 = 0, NCCL_INT32 = 2, NCCL_FLOAT64 = 8, NCCL_SUM = 0 };

struct RcclFns { nccl_allreduce_t allreduce; nccl_allgather_t allgather; nccl_send_t send; nccl_recv_t recv; nccl_group_t gstart, gend; nccl_bcast_t bcast; };
static RcclFns g_rccl;

static void *open_rccl()
{
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  return h;
}

extern "C" int ks_comm_get_unique_id(unsigned char id[KS_UNIQUE_ID_BYTES])
{
  KS_CHECK(id, KS_ERR_ARG_NULL, "id is NULL");
  void *h = open_rccl();
  KS_CHECK(h, KS_ERR_LIB, "cannot open librccl: %s", dlerror());
  nccl_getuid_t f = (nccl_getuid_t)dlsym(h, "ncclGetUniqueId");
  KS_CHECK(f, KS_ERR_LIB, "ncclGetUniqueId not found");
  NcclUniqueIdRaw raw; memset(&raw, 0, sizeof(raw));
  int rc = f(&raw);
  KS_CHECK(rc == 0, KS_ERR_LIB, "ncclGetUniqueId failed (%d)", rc);
  memcpy(id, raw.internal, KS_UNIQUE_ID_BYTES);
  return KS_SUCCESS;
}

// RCCL implementations of the three provider operations; `user` is the ks_ctx
static int rccl_allreduce_sum(void *user, double *dev_buf, int count, void *stream)
{
  ks_ctx ctx = (ks_ctx)user;
  return g_rccl.allreduce(dev_buf, dev_buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM, ctx->comm.nccl_comm, (hipStream_t)stream);
}
static int rccl_allgather_host(void *user, const void *send, int bytes, void *recv)
{
  ks_ctx ctx = (ks_ctx)user;
  const size_t need = (size_t)bytes * (ctx->comm.size + 1);
  if (ctx->comm.ag_len < need) {
    if (ctx->comm.ag_dev) hipFree(ctx->comm.ag_dev);
    ctx->comm.ag_dev = nullptr; ctx->comm.ag_len = 0;
    if (hipMalloc(&ctx->comm.ag_dev, need) != hipSuccess) return 1;
    ctx->comm.ag_len = need;
  }
  char *d = ctx->comm.ag_dev;
  int rc = 0;
  if (hipMemcpyAsync(d + (size_t)bytes * ctx->comm.size, send, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 2;
  if (!rc) rc = g_rccl.allgather(d + (size_t)bytes * ctx->comm.size, d, (size_t)bytes, NCCL_INT8, ctx->comm.nccl_comm, ctx->stream);
  if (!rc && hipMemcpyAsync(recv, d, (size_t)bytes * ctx->comm.size, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 3;
  if (ks_sync(ctx) != hipSuccess && !rc) rc = 4;
  return rc;
}
static int rccl_exchange(void *user, int npeers, const int *peers, const void *dev_send, const int *send_off, const int *send_cnt,
                         void *dev_recv, const int *recv_off, const int *recv_cnt, int elem_bytes, void *stream)
{
  ks_ctx ctx = (ks_ctx)user;
  const int dt = elem_bytes == 8 ? NCCL_FLOAT64 : (elem_bytes == 4 ? NCCL_INT32 : NCCL_INT8);
  const size_t mul = (dt == NCCL_INT8) ? (size_t)elem_bytes : 1;
  int rc = g_rccl.gstart();
  for (int i = 0; i < npeers && !rc; i++) {
    if (send_cnt[i]) rc = g_rccl.send((const char *)dev_send + (size_t)send_off[i] * elem_bytes, (size_t)send_cnt[i] * mul, dt, peers[i], ctx->comm.nccl_comm, (hipStream_t)stream);
    if (!rc && recv_cnt[i]) rc = g_rccl.recv((char *)dev_recv + (size_t)recv_off[i] * elem_bytes, (size_t)recv_cnt[i] * mul, dt, peers[i], ctx->comm.nccl_comm, (hipStream_t)stream);
  }
  int rc2 = g_rccl.gend();
  return rc ? rc : rc2;
}

extern "C" int ks_comm_init_rccl(ks_ctx ctx, int rank, int size, const unsigned char id[KS_UNIQUE_ID_BYTES])
{
  KS_CHECK(ctx && id, KS_ERR_ARG_NULL, "ctx or id is NULL");
  KS_CHECK(size >= 1 && rank >= 0 && rank < size, KS_ERR_ARG_OUTOFRANGE, "bad rank/size %d/%d", rank, size);
  KS_HIP(hipSetDevice(ctx->device));
  void *h = open_rccl();
  KS_CHECK(h, KS_ERR_LIB, "cannot open librccl: %s", dlerror());
  nccl_initrank_t f = (nccl_initrank_t)dlsym(h, "ncclCommInitRank");
  g_rccl.allreduce = (nccl_allreduce_t)dlsym(h, "ncclAllReduce");
  g_rccl.allgather = (nccl_allgather_t)dlsym(h, "ncclAllGather");
  g_rccl.send = (nccl_send_t)dlsym(h, "ncclSend");
  g_rccl.recv = (nccl_recv_t)dlsym(h, "ncclRecv");
  g_rccl.gstart = (nccl_group_t)dlsym(h, "ncclGroupStart");
  g_rccl.gend = (nccl_group_t)dlsym(h, "ncclGroupEnd");
  g_rccl.bcast = (nccl_bcast_t)dlsym(h, "ncclBroadcast");        // optional: without it the broadcast of the projected problem goes through the allgather
  KS_CHECK(f && g_rccl.allreduce && g_rccl.allgather && g_rccl.send && g_rccl.recv && g_rccl.gstart && g_rccl.gend, KS_ERR_LIB, "RCCL symbols not found");
  NcclUniqueIdRaw raw; memcpy(raw.internal, id, KS_UNIQUE_ID_BYTES);
  void *comm = nullptr;
  int rc = f(&comm, size, raw, rank);
  KS_CHECK(rc == 0, KS_ERR_LIB, "ncclCommInitRank failed (%d)", rc);
  ctx->comm.rccl_lib = h; ctx->comm.nccl_comm = comm; ctx->comm.rank = rank; ctx->comm.size = size;
  ctx->comm.ops.allreduce_sum = rccl_allreduce_sum; ctx->comm.ops.allgather_host = rccl_allgather_host; ctx->comm.ops.exchange = rccl_exchange;
  ctx->comm.user = ctx;
  ctx->comm.force_collectives = ctx->dbg.force_multi;
  return KS_SUCCESS;
}

extern "C" int ks_comm_set_ops(ks_ctx ctx, int rank, int size, const ks_comm_ops *ops, void *user)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  KS_CHECK(size >= 1 && rank >= 0 && rank < size, KS_ERR_ARG_OUTOFRANGE, "bad rank/size %d/%d", rank, size);
  KS_CHECK(size == 1 || (ops && ops->allreduce_sum && ops->allgather_host && ops->exchange), KS_ERR_ARG_NULL, "all three operations are required when size>1");
  ctx->comm.rank = rank; ctx->comm.size = size;
  if (ops) ctx->comm.ops = *ops;
  ctx->comm.user = user;
  ctx->comm.force_collectives = ops && ops->allreduce_sum && ctx->dbg.force_multi;
  return KS_SUCCESS;
}

extern "C" int ks_comm_rank_size(ks_ctx ctx, int *rank, int *size)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  if (rank) *rank = ctx->comm.rank;
  if (size) *size = ctx->comm.size;
  return KS_SUCCESS;
}

extern "C" int ks_ctx_memcpy_stream(ks_ctx ctx, void *dst, const void *src, size_t bytes, int kind, void *stream)
{
  KS_CHECK(ctx && (bytes == 0 || (dst && src)), KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  if (bytes) KS_HIP(hipMemcpyAsync(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, s));
  KS_HIP(hipStreamSynchronize(s));
  return KS_SUCCESS;
}

extern "C" int ks_ctx_memcpy(ks_ctx ctx, void *dst, const void *src, size_t bytes, int kind)
{
  KS_CHECK(ctx && (bytes == 0 || (dst && src)), KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(ctx->device));
  if (bytes) KS_HIP(hipMemcpyAsync(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(ks_sync(ctx));
  return KS_SUCCESS;
}

extern "C" int ks_ctx_memset(ks_ctx ctx, void *dev, int value, size_t bytes)
{
  KS_CHECK(ctx && (bytes == 0 || dev), KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(ctx->device));
  if (bytes) KS_HIP(hipMemsetAsync(dev, value, bytes, ctx->stream));
  return KS_SUCCESS;
}

// ---- one-shot allreduce -------------------------------------------------------------------------
// the k+1 <= 61 doubles of a Gram-Schmidt pass are summed by ONE kernel per rank and no library call: every rank
// writes its values into a mailbox of every rank (its own included) and then adds what arrived in its own mailbox in rank
// order, so all ranks hold the same bits. A double travels as two 8-byte packets {sequence number, 32 data bits}: an
// 8-byte store is indivisible on the fabric, so a packet whose sequence number matches carries its data and no fence or
// separate flag is needed. Mailboxes live in uncached device memory (remote stores land in memory, local polls read memory).
// Two parities of slots: a rank can be at most one call ahead of another (it needs that rank's packets of call s+1 to get
// past s+1, and those are sent after the rank has consumed call s), so packets of call s+2 never overwrite unread ones of s.
// Every wait is bounded: a rank that sees no packet for timeout_ticks gives up, poisons its result with NaN and raises the
// error word the host finds at its next wait.
__global__ __launch_bounds__(512) void k_allreduce_oneshot(double *__restrict__ buf, int count, KsOneShotArgs o)
{
  __shared__ unsigned sh[KS_ONESHOT_MAX_RANKS][2 * KS_ONESHOT_MAX_COUNT];
  __shared__ int failed;
  ks_oneshot_sum(buf, buf, count, o, sh, &failed);
}

static int oneshot_allreduce(ks_ctx ctx, double *dev_buf, int count)
{
  auto &os = ctx->comm.oneshot;
  KsOneShotArgs o;
  KS_CHECK(ks_oneshot_next(ctx, count, &o), KS_ERR_PLIB, "one-shot allreduce is not active");
  (void)os;
  hipLaunchKernelGGL(k_allreduce_oneshot, dim3(1), dim3(512), 0, ctx->stream, dev_buf, count, o);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

// the arguments of the next one-shot call (its sequence number is consumed); false: not active, or too long a message
bool ks_oneshot_next(ks_ctx ctx, int count, KsOneShotArgs *o)
{
  auto &os = ctx->comm.oneshot;
  if (!os.enabled || count > KS_ONESHOT_MAX_COUNT || count <= 0) return false;
  os.par ^= 1u;                                      // the two-parity argument above needs a flip on every call, also across the stamp's wrap
  if (++os.seq == 0) ++os.seq;                       // 0 is what an untouched mailbox holds
  for (int r = 0; r < KS_ONESHOT_MAX_RANKS; r++) o->peer[r] = os.peer[r];
  o->mine = os.mine; o->err = os.err_dev; o->err_local = os.err_local; o->timeout_ticks = os.timeout_ticks; o->seq = os.seq; o->par = os.par; o->me = ctx->comm.rank; o->size = ctx->comm.size;
  return true;
}

int ks_oneshot_error(ks_ctx ctx)
{
  const volatile int *e = ctx->comm.oneshot.err_host;
  if (e && *e) KS_FAIL(KS_ERR_LIB, "one-shot allreduce number %d saw no packet from some rank within its time limit (results since then are NaN)", *e);
  const volatile int *hh = ctx->comm.halo_err_host;
  if (hh && *hh) KS_FAIL(KS_ERR_LIB, "a peer-mapped halo exchange gave up waiting for a neighbour (ghost values since then are NaN)");
  return KS_SUCCESS;
}

void ks_oneshot_release(ks_ctx ctx)
{
  auto &os = ctx->comm.oneshot;
  os.enabled = false;
  for (int r = 0; r < KS_ONESHOT_MAX_RANKS; r++) {
    if (os.opened[r] && os.peer[r]) hipIpcCloseMemHandle(os.peer[r]);
    os.peer[r] = nullptr; os.opened[r] = false;
  }
  if (os.mine) hipFree(os.mine);
  os.mine = nullptr;
  if (os.err_host) hipHostFree(os.err_host);
  os.err_host = os.err_dev = nullptr;
  if (os.err_local) hipFree(os.err_local);
  os.err_local = nullptr;
  (void)hipGetLastError();
}

struct OneShotHello { int ok, pid, device, pad; unsigned long long ptr; hipIpcMemHandle_t handle; };

// Collective. kind KS_ALLREDUCE_ONESHOT: map every rank's mailbox; the one-shot path is switched on only if EVERY rank managed
// (the ranks agree through a second allgather), otherwise everything is released again and the provider's allreduce stays.
extern "C" int ks_comm_set_allreduce(ks_ctx ctx, int kind, int *active)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  KS_CHECK(kind == KS_ALLREDUCE_PROVIDER || kind == KS_ALLREDUCE_ONESHOT, KS_ERR_ARG_OUTOFRANGE, "unknown allreduce kind %d", kind);
  KS_HIP(hipSetDevice(ctx->device));
  KS_HIP(hipStreamSynchronize(ctx->stream));
  auto &os = ctx->comm.oneshot;
  if (active) *active = KS_ALLREDUCE_PROVIDER;
  ks_oneshot_release(ctx);
  if (kind == KS_ALLREDUCE_PROVIDER || !ks_is_multi(ctx)) return KS_SUCCESS;
  const int size = ctx->comm.size, rank = ctx->comm.rank;
  KS_CHECK(ctx->comm.ops.allgather_host, KS_ERR_ORDER, "no communicator installed");
  const size_t bytes = (size_t)2 * KS_ONESHOT_MAX_RANKS * 2 * KS_ONESHOT_MAX_COUNT * sizeof(unsigned long long);
  OneShotHello me; memset(&me, 0, sizeof(me));
  me.pid = (int)getpid(); me.device = ctx->device;
  bool ok = size <= KS_ONESHOT_MAX_RANKS;
  if (ok) {
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    if (!p && hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    os.mine = (unsigned long long *)p;
    ok = p && hipMemset(p, 0, bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (ok) ok = hipHostMalloc((void **)&os.err_host, sizeof(int), hipHostMallocMapped) == hipSuccess;
    if (ok) { *os.err_host = 0; ok = hipHostGetDevicePointer((void **)&os.err_dev, os.err_host, 0) == hipSuccess; }
    if (ok) ok = hipMalloc((void **)&os.err_local, sizeof(int)) == hipSuccess && hipMemset(os.err_local, 0, sizeof(int)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (ok) {
      me.ptr = (unsigned long long)(uintptr_t)p;
      if (hipIpcGetMemHandle(&me.handle, p) != hipSuccess) { (void)hipGetLastError(); me.pad = 1; }     // pad = 1: no handle; ranks of this process can still use the pointer
    }
  }
  me.ok = ok ? 1 : 0;
  std::vector<OneShotHello> all(size);
  KS_CALL(ks_comm_allgather_host(ctx, &me, (int)sizeof(me), all.data()));
  for (int r = 0; r < size && ok; r++) {
    if (!all[r].ok) { ok = false; break; }
    if (r == rank) { os.peer[r] = os.mine; continue; }
    if (all[r].pid == me.pid) {
      if (all[r].device != ctx->device) {
        hipError_t e = hipDeviceEnablePeerAccess(all[r].device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
        (void)hipGetLastError();
      }
      os.peer[r] = (unsigned long long *)(uintptr_t)all[r].ptr;
    } else {
      void *q = nullptr;
      if (all[r].pad || hipIpcOpenMemHandle(&q, all[r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = false; }
      else { os.peer[r] = (unsigned long long *)q; os.opened[r] = true; }
    }
  }
  int mine_ok = ok ? 1 : 0;
  std::vector<int> oks(size, 0);
  KS_CALL(ks_comm_allgather_host(ctx, &mine_ok, (int)sizeof(int), oks.data()));
  for (int r = 0; r < size; r++) if (!oks[r]) ok = false;
  if (!ok) { ks_oneshot_release(ctx); return KS_SUCCESS; }
  int khz = 100000;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
  const char *tm = getenv("KSGPU_ONESHOT_TIMEOUT_MS");
  const long long ms = tm && atoll(tm) > 0 ? atoll(tm) : 2000;
  os.timeout_ticks = ms * khz;
  os.seq = ctx->dbg.oneshot_seq0;                              // test hook (KS_DEBUG_ONESHOT_SEQ0): start the stamps near their 32-bit wrap
  os.par = 0;
  os.enabled = true;
  if (active) *active = KS_ALLREDUCE_ONESHOT;
  return KS_SUCCESS;
}

extern "C" int ks_comm_allreduce_sum(ks_ctx ctx, double *dev_buf, int count)
{
  KS_CHECK(ctx && dev_buf, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(count >= 0, KS_ERR_ARG_OUTOFRANGE, "negative count");
  KS_HIP(hipSetDevice(ctx->device));
  return ks_allreduce_sum(ctx, dev_buf, count);
}

extern "C" int ks_comm_get_allreduce(ks_ctx ctx, int *active)
{
  KS_CHECK(ctx && active, KS_ERR_ARG_NULL, "NULL argument");
  *active = ctx->comm.oneshot.enabled ? KS_ALLREDUCE_ONESHOT : KS_ALLREDUCE_PROVIDER;
  return KS_SUCCESS;
}

// In-place SUM allreduce of `count` doubles in device memory, stream-ordered (bvblas.c:255 MPIU_Allreduce).
int ks_allreduce_sum(ks_ctx ctx, double *dev_buf, int count)
{
  if (!ks_is_multi(ctx) || count <= 0) return KS_SUCCESS;
  KS_CHECK(ctx->comm.ops.allreduce_sum, KS_ERR_ORDER, "size>1 but no communicator: call ks_comm_init_rccl or ks_comm_set_ops");
  KsProfScope ps(ctx, KS_K_ALLREDUCE, 8.0 * count);
  if (ctx->comm.oneshot.enabled && count <= KS_ONESHOT_MAX_COUNT) return oneshot_allreduce(ctx, dev_buf, count);
  int rc = ctx->comm.ops.allreduce_sum(ctx->comm.user, dev_buf, count, (void *)ctx->stream);
  KS_CHECK(rc == 0, KS_ERR_LIB, "allreduce failed (%d)", rc);
  return KS_SUCCESS;
}

int ks_comm_allgather_host(ks_ctx ctx, const void *send, int bytes, void *recv)
{
  if (ctx->comm.size == 1 && !(ctx->comm.force_collectives && ctx->comm.ops.allgather_host)) { memcpy(recv, send, bytes); return KS_SUCCESS; }
  KS_CHECK(ctx->comm.ops.allgather_host, KS_ERR_ORDER, "no communicator");
  int rc = ctx->comm.ops.allgather_host(ctx->comm.user, send, bytes, recv);
  KS_CHECK(rc == 0, KS_ERR_LIB, "allgather failed (%d)", rc);
  return KS_SUCCESS;
}

extern "C" int ks_comm_bcast_stats(ks_ctx ctx, long long *calls, double *seconds, int reset)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  if (calls) *calls = ctx->comm.bcast_calls;
  if (seconds) *seconds = ctx->comm.bcast_seconds;
  if (reset) { ctx->comm.bcast_calls = 0; ctx->comm.bcast_seconds = 0.0; }
  return KS_SUCCESS;
}

static int bcast0_host(ks_ctx ctx, void *buf, int bytes);
int ks_comm_bcast0_host(ks_ctx ctx, void *buf, int bytes)
{
  if ((ctx->comm.size == 1 && !ctx->comm.force_collectives) || bytes <= 0) return KS_SUCCESS;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = bcast0_host(ctx, buf, bytes);
  ctx->comm.bcast_calls++;
  ctx->comm.bcast_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
static int bcast0_host(ks_ctx ctx, void *buf, int bytes)
{
  if (ctx->comm.nccl_comm && g_rccl.bcast && ctx->comm.ops.allgather_host == rccl_allgather_host) {
    // native provider: ONE ncclBroadcast of rank 0's bytes through the device staging area and one host wait (the allgather form below
    // ships size x bytes and is what a caller-supplied provider, which has no broadcast slot, is left with)
    if (ctx->comm.ag_len < (size_t)bytes) {
      if (ctx->comm.ag_dev) hipFree(ctx->comm.ag_dev);
      ctx->comm.ag_dev = nullptr; ctx->comm.ag_len = 0;
      KS_HIP(hipMalloc(&ctx->comm.ag_dev, (size_t)bytes));
      ctx->comm.ag_len = (size_t)bytes;
    }
    if (ctx->comm.rank == 0) KS_HIP(hipMemcpyAsync(ctx->comm.ag_dev, buf, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    const int rc = g_rccl.bcast(ctx->comm.ag_dev, ctx->comm.ag_dev, (size_t)bytes, NCCL_INT8, 0, ctx->comm.nccl_comm, ctx->stream);
    KS_CHECK(rc == 0, KS_ERR_LIB, "ncclBroadcast failed (%d)", rc);
    if (ctx->comm.rank != 0) KS_HIP(hipMemcpyAsync(buf, ctx->comm.ag_dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
    return KS_SUCCESS;
  }
  std::vector<char> all((size_t)bytes * ctx->comm.size);
  KS_CALL(ks_comm_allgather_host(ctx, buf, bytes, all.data()));
  if (ctx->comm.rank) memcpy(buf, all.data(), (size_t)bytes);
  return KS_SUCCESS;
}

int ks_comm_exchange(ks_ctx ctx, int npeers, const int *peers, const void *dev_send, const int *send_off, const int *send_cnt,
                     void *dev_recv, const int *recv_off, const int *recv_cnt, int elem_bytes, hipStream_t stream)
{
  if (npeers == 0) return KS_SUCCESS;
  KS_CHECK(ctx->comm.ops.exchange, KS_ERR_ORDER, "no communicator");
  int rc = ctx->comm.ops.exchange(ctx->comm.user, npeers, peers, dev_send, send_off, send_cnt, dev_recv, recv_off, recv_cnt, elem_bytes, (void *)(stream ? stream : ctx->stream));
  KS_CHECK(rc == 0, KS_ERR_LIB, "neighbour exchange failed (%d)", rc);
  return KS_SUCCESS;
}

// Known-answer run of the three provider operations (what an integrator calls once after installing a communicator, and
// what bench.py calls before the timed region): allreduces whose sums are known in closed form (64 of them enqueued back to
// back, so that a one-shot path goes through its slot parities without a host wait in between), allgathers of one int per
// rank, and a ring exchange with the neighbours rank+1 and rank-1 (with itself at size 1 under the force_multi test hook).
// Every rank runs every stage whatever it has seen so far, and the verdict is agreed through a last allgather: the call
// returns the same on all ranks and leaves no rank waiting in a collective the others skipped.
extern "C" int ks_comm_check(ks_ctx ctx)
{
  KS_CHECK(ctx, KS_ERR_ARG_NULL, "ctx is NULL");
  KS_HIP(hipSetDevice(ctx->device));
  const int size = ctx->comm.size, rank = ctx->comm.rank;
  if (!ks_is_multi(ctx)) return KS_SUCCESS;
  KS_CHECK(ctx->comm.ops.allreduce_sum && ctx->comm.ops.allgather_host && ctx->comm.ops.exchange, KS_ERR_ORDER, "no communicator installed");
  constexpr int NCALL = 64, LEN = 8;
  double *d = nullptr;
  std::vector<double> h(NCALL * LEN + 16);
  bool bad = false;
  char msg[400] = "";
  auto fail = [&](const char *fmt, auto... a) {
    if (!bad) { if constexpr (sizeof...(a) == 0) snprintf(msg, sizeof(msg), "%s", fmt); else snprintf(msg, sizeof(msg), fmt, a...); }
    bad = true;
  };
  // (0) a rank whose own set-up fails must not leave the others waiting in the stages below: agree on the set-up first
  {
    if (hipMalloc(&d, (NCALL * LEN + 16) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
    std::vector<int> oks(size, 0);
    const int mine_ok = d ? 1 : 0;
    if (ctx->comm.ops.allgather_host(ctx->comm.user, &mine_ok, (int)sizeof(int), oks.data()) != 0) { if (d) hipFree(d); KS_FAIL(KS_ERR_LIB, "communicator check: allgather failed"); }
    for (int r = 0; r < size; r++) if (!oks[r]) { if (d) hipFree(d); KS_FAIL(KS_ERR_MEM, "communicator check: rank %d could not allocate its test buffer", r); }
  }
  // (1) allreduce
  for (int i = 0; i < NCALL; i++) for (int j = 0; j < LEN; j++) h[i * LEN + j] = (rank + 1) * (i + 1) * (j + 1) / 16.0;
  if (hipMemcpyAsync(d, h.data(), NCALL * LEN * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) fail("communicator check: upload failed");
  for (int i = 0; i < NCALL; i++)
    if (ks_allreduce_sum(ctx, d + i * LEN, LEN) != KS_SUCCESS) fail("communicator check: allreduce call %d failed: %s", i, ks_last_error_message());
  if (hipMemcpyAsync(h.data(), d, NCALL * LEN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || ks_sync(ctx) != hipSuccess) {
    if (ks_oneshot_error(ctx)) fail("communicator check: %s", ks_last_error_message());        // the one-shot path gave up waiting
    else fail("communicator check: waiting for the allreduces failed");
  }
  for (int i = 0; i < NCALL && !bad; i++) for (int j = 0; j < LEN; j++) {
    const double want = 0.5 * size * (size + 1) * (i + 1) * (j + 1) / 16.0;
    if (h[i * LEN + j] != want) { fail("communicator check: allreduce %d gave %g for entry %d, expected %g", i, h[i * LEN + j], j, want); break; }
  }
  // (2) allgather, twice: the staging buffer is reused
  for (int rep = 0; rep < 2; rep++) {
    std::vector<int> all(size, -1);
    const int mine = 7 * rank + 1 + rep;
    if (ctx->comm.ops.allgather_host(ctx->comm.user, &mine, (int)sizeof(int), all.data()) != 0) fail("communicator check: allgather failed");
    for (int r = 0; r < size; r++) if (all[r] != 7 * r + 1 + rep) fail("communicator check: allgather slot %d holds %d, expected %d", r, all[r], 7 * r + 1 + rep);
  }
  // (3) ring exchange
  if (size > 1 || ctx->comm.nccl_comm) {                  // a caller-supplied provider need not know how to exchange with itself
    int peers[2] = { (rank + 1) % size, (rank + size - 1) % size };
    const int np = peers[0] == peers[1] ? 1 : 2;
    int soff[2] = { 0, 2 }, roff[2] = { 0, 2 }, cnt[2] = { 2, 2 };
    for (int i = 0; i < np; i++) { h[2 * i] = 100.0 * rank + peers[i]; h[2 * i + 1] = -h[2 * i]; }
    for (int i = 4; i < 8; i++) h[i] = 0.0;
    if (hipMemcpyAsync(d, h.data(), 8 * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) fail("communicator check: upload failed");
    if (ctx->comm.ops.exchange(ctx->comm.user, np, peers, d, soff, cnt, d + 4, roff, cnt, (int)sizeof(double), (void *)ctx->stream) != 0) fail("communicator check: neighbour exchange failed");
    if (hipMemcpyAsync(h.data(), d, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) fail("communicator check: waiting for the exchange failed");
    for (int i = 0; i < np; i++) {
      const double want = 100.0 * peers[i] + rank;
      if (h[4 + 2 * i] != want || h[4 + 2 * i + 1] != -want) fail("communicator check: exchange with rank %d delivered %g %g, expected %g %g", peers[i], h[4 + 2 * i], h[4 + 2 * i + 1], want, -want);
    }
  }
  // (4) the verdict, agreed
  std::vector<int> verdict(size, 1);
  const int mine = bad ? 1 : 0;
  (void)ctx->comm.ops.allgather_host(ctx->comm.user, &mine, (int)sizeof(int), verdict.data());
  hipFree(d);
  (void)hipGetLastError();
  if (bad) KS_FAIL(KS_ERR_LIB, "%s", msg);
  for (int r = 0; r < size; r++) if (verdict[r]) KS_FAIL(KS_ERR_LIB, "communicator check: rank %d reported a failure", r);
  return KS_SUCCESS;
}
