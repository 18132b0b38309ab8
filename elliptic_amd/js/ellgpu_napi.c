/*
 * ellgpu_napi.c -- Node N-API addon: the JavaScript-side binding of the C ABI
 * in include/ellgpu.h.  It marshals flat Buffers (fixed-width big-endian
 * integers, exactly BN#toArray('be', len)) into libellgpu.so and back; no curve
 * arithmetic lives here.  The library is dlopen()ed from an explicit path
 * (default ../lib/libellgpu.so next to this addon), so the same addon can be
 * pointed at the CPU unit-test build of the device code in a GPU-less
 * container (tests only, see tests/test_js_install.py).
 *
 *   gcc -shared -fPIC -I/usr/include/node -o ellgpu.node ellgpu_napi.c -ldl
 *
 * Exports:  open(path) -> true
 *           createContext(device) -> external
 *           destroyContext(ctx)
 *           combBits(ctx, curve) -> width of the fixed-base table in use (ellgpu_ctx_comb_bits)
 *           defer(ctx): the next few-item call on ctx returns once it is enqueued (ellgpu_ctx_defer);
 *           collect(ctx): waits for it and fills the result Buffers that call returned
 *             (ellgpu_ctx_collect; throws the library's error) -- index.js validates the
 *             reference's precomputed tables in between, while the device works
 *           curveId(name), fieldBytes(id), orderBytes(id), deviceCount(), defineShort(ctx, p, a, b), defineEdwards(ctx, p, a, d)
 *           mulFixed(ctx, curve, k) -> {xy, inf}
 *           mulVar(ctx, curve, k, xy) -> {xy, inf}
 *           mulAdd2(ctx, curve, k1, p1|null, k2, p2) -> {xy, inf}
 *           ecdsaVerify(ctx, curve, hash, hashLen, msgBits, r, s, pub) -> {ok, status}
 *           x25519(ctx, k, x) -> {x, inf}
 *           x25519Derive(ctx, k, x) -> {x, status}   (KeyPair#derive: validate + ladder, one call)
 *           decompress(ctx, curve, v, odd) -> {xy, ok}
 *           ecdsaSign(ctx, curve, hash, hashLen, msgBits, priv, nonces, canonical)
 *             -> {r, s, recid, ok}
 *           eddsaVerify(ctx, msgs, offsets|null, msgLen, sigs, pubs) -> {ok, err}
 *           eddsaSign(ctx, msgs, offsets|null, msgLen, secrets) -> {sig, pub}
 *           ecdsaRecover(ctx, curve, hash, hashLen, r, s, recid) -> {xy, status}
 *           ecdsaSignDet(ctx, curve, hash, hashLen, msgBits, priv, canonical) -> {r, s, recid, ok}
 *           decodePoints(ctx, curve, enc, encLen) -> {xy, status}
 *           encodePoints(ctx, curve, xy, compact) -> Buffer
 *           validate(ctx, curve, xy, inf|null, checkOrder) -> Buffer (status bytes)
 *           pointAdd(ctx, curve, xy1, inf1|null, xy2, inf2|null) -> {xy, inf}
 *           sigFromDer(ctx, curve, der, stride, lens) -> {r, s, status}
 *           sigToDer(ctx, curve, r, s) -> {der, lens}     (stride = der.length / n)
 *           ecdsaVerifyWire(ctx, curve, hash, hashLen, msgBits, der, stride, lens, keys, keyLen)
 *             -> {ok, err};  lens: Buffer of n little-endian uint32
 *             offsets: Buffer of n+1 little-endian uint64 byte offsets into msgs
 *           callAsync(op, ctx, curve, hashLen, msgBits, b0, b1, b2, b3) -> Promise
 *             op 0 mulFixed(b0=k) 1 mulVar(k, xy) 2 mulAdd2(k1, p1|null, k2, p2)
 *             3 ecdsaVerify(hash, r, s, pub) 4 x25519(k, x) 5 ecdsaSignDet(hash, priv; i0 = canonical)
 *             6 ecdsaRecover(hash, r, s, recid) 7 ecdsaVerifyWire(hash, der, lens, keys; i0 = der
 *             stride, i1 = key length) 8 decodePoints(enc; i0 = encoding length); runs on a libuv worker
 *             thread (napi_async_work) so the JS thread is not blocked; resolves to the
 *             same value the synchronous form returns.  One call per context at a time:
 *             index.js serialises them.
 * Errors from the library are thrown as plain `Error(message)`, the
 * reference's convention (minimalistic-assert, dist/elliptic.js:8832-8835).
 */
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

typedef struct ellgpu_ctx ellgpu_ctx;
/* ELLGPU_VERSION of the include/ellgpu.h this file was written against: open() refuses a library
 * that reports another one (a prototype changed; dlsym would still find the name) */
#define ELLGPU_ABI_VERSION 0x000200
static struct {
  void* h;
  int (*version)(void);
  int (*ctx_defer)(ellgpu_ctx*);
  int (*ctx_comb_bits)(ellgpu_ctx*, int);
  int (*ctx_collect)(ellgpu_ctx*);
  const char* (*last_error)(void);
  int (*curve_id)(const char*);
  int (*field_bytes)(int);
  int (*order_bytes)(int);
  int (*device_count)(void);
  int (*ctx_create)(int, ellgpu_ctx**);
  int (*group_create)(const int*, int, ellgpu_ctx**);
  int (*group_size)(const ellgpu_ctx*);
  int (*define_short)(ellgpu_ctx*, const uint8_t*, const uint8_t*, const uint8_t*, int*);
  int (*define_edwards)(ellgpu_ctx*, const uint8_t*, const uint8_t*, const uint8_t*, int*);
  void (*ctx_destroy)(ellgpu_ctx*);
  int (*mul_fixed)(ellgpu_ctx*, int, size_t, const uint8_t*, uint8_t*, uint8_t*);
  int (*mul_var)(ellgpu_ctx*, int, size_t, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
  int (*mul_add2)(ellgpu_ctx*, int, size_t, const uint8_t*, const uint8_t*, const uint8_t*,
                  const uint8_t*, uint8_t*, uint8_t*);
  int (*ecdsa_verify)(ellgpu_ctx*, int, size_t, const uint8_t*, int, int, const uint8_t*,
                      const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
  int (*x25519)(ellgpu_ctx*, size_t, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
  int (*x25519_derive)(ellgpu_ctx*, size_t, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
  int (*decompress)(ellgpu_ctx*, int, size_t, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
  int (*ecdsa_sign)(ellgpu_ctx*, int, size_t, const uint8_t*, int, int, const uint8_t*, const uint8_t*,
                    int, uint8_t*, uint8_t*, uint8_t*, uint8_t*);
  int (*eddsa_verify)(ellgpu_ctx*, size_t, const uint8_t*, const uint64_t*, size_t, const uint8_t*,
                      const uint8_t*, uint8_t*, uint8_t*);
  int (*eddsa_sign)(ellgpu_ctx*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, size_t,
                    uint8_t*, uint8_t*);
  int (*ecdsa_recover)(ellgpu_ctx*, int, size_t, const uint8_t*, int, const uint8_t*, const uint8_t*,
                       const uint8_t*, uint8_t*, uint8_t*);
  int (*ecdsa_sign_det)(ellgpu_ctx*, int, size_t, const uint8_t*, int, int, const uint8_t*, int,
                        uint8_t*, uint8_t*, uint8_t*, uint8_t*);
  int (*decode_points)(ellgpu_ctx*, int, size_t, const uint8_t*, size_t, uint8_t*, uint8_t*);
  int (*encode_points)(ellgpu_ctx*, int, size_t, const uint8_t*, int, uint8_t*);
  int (*validate)(ellgpu_ctx*, int, size_t, const uint8_t*, const uint8_t*, int, uint8_t*);
  int (*point_add)(ellgpu_ctx*, int, size_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*,
                   uint8_t*, uint8_t*);
  int (*sig_from_der)(ellgpu_ctx*, int, size_t, const uint8_t*, size_t, const uint32_t*, uint8_t*, uint8_t*,
                      uint8_t*);
  int (*sig_to_der)(ellgpu_ctx*, int, size_t, const uint8_t*, const uint8_t*, uint8_t*, size_t, uint32_t*);
  int (*verify_wire)(ellgpu_ctx*, int, size_t, const uint8_t*, int, int, const uint8_t*, size_t,
                     const uint32_t*, const uint8_t*, size_t, uint8_t*, uint8_t*);
} L;

#define THROW(env, msg) do { napi_throw_error((env), NULL, (msg)); return NULL; } while (0)
#define CHECK(env, call) do { if ((call) != napi_ok) THROW(env, "N-API call failed"); } while (0)

static napi_value lib_error(napi_env env) {
  const char* m = L.last_error ? L.last_error() : "ellgpu error";
  napi_throw_error(env, NULL, m && *m ? m : "ellgpu error");
  return NULL;
}
static int need_lib(napi_env env) {
  if (!L.h) { napi_throw_error(env, NULL, "ellgpu: library not opened (call open(path) first)"); return 0; }
  return 1;
}

static napi_value fn_open(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  char path[4096]; size_t len = 0;
  if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof path, &len) != napi_ok)
    THROW(env, "open(path): path string required");
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { char msg[4400]; snprintf(msg, sizeof msg, "ellgpu: cannot load %s: %s", path, dlerror()); THROW(env, msg); }
#define SYM(field, name) do { *(void**)(&L.field) = dlsym(h, name); if (!L.field) { char m_[256]; \
    snprintf(m_, sizeof m_, "ellgpu: %s missing from library", name); dlclose(h); THROW(env, m_); } } while (0)
  SYM(version, "ellgpu_version");
  if (L.version() != ELLGPU_ABI_VERSION) {
    char m_[256];
    snprintf(m_, sizeof m_, "ellgpu: library reports ABI version 0x%06x, this addon was built for 0x%06x (include/ellgpu.h)",
             L.version(), ELLGPU_ABI_VERSION);
    dlclose(h); THROW(env, m_);
  }
  SYM(ctx_defer, "ellgpu_ctx_defer"); SYM(ctx_collect, "ellgpu_ctx_collect");
  SYM(ctx_comb_bits, "ellgpu_ctx_comb_bits");
  SYM(last_error, "ellgpu_last_error"); SYM(curve_id, "ellgpu_curve_id");
  SYM(field_bytes, "ellgpu_curve_field_bytes"); SYM(order_bytes, "ellgpu_curve_order_bytes");
  SYM(device_count, "ellgpu_device_count"); SYM(ctx_create, "ellgpu_ctx_create");
  SYM(group_create, "ellgpu_group_create"); SYM(group_size, "ellgpu_group_size");
  SYM(define_short, "ellgpu_curve_define_short");
  SYM(define_edwards, "ellgpu_curve_define_edwards");
  SYM(ctx_destroy, "ellgpu_ctx_destroy"); SYM(mul_fixed, "ellgpu_mul_fixed"); SYM(mul_var, "ellgpu_mul_var");
  SYM(mul_add2, "ellgpu_mul_add2"); SYM(ecdsa_verify, "ellgpu_ecdsa_verify"); SYM(x25519, "ellgpu_x25519_ladder");
  SYM(x25519_derive, "ellgpu_x25519_derive");
  SYM(decompress, "ellgpu_decompress");
  SYM(eddsa_verify, "ellgpu_eddsa_verify");
  SYM(eddsa_sign, "ellgpu_eddsa_sign");
  SYM(ecdsa_recover, "ellgpu_ecdsa_recover");
  SYM(ecdsa_sign_det, "ellgpu_ecdsa_sign_det");
  SYM(ecdsa_sign, "ellgpu_ecdsa_sign");
  SYM(decode_points, "ellgpu_decode_points");
  SYM(encode_points, "ellgpu_encode_points");
  SYM(validate, "ellgpu_validate");
  SYM(point_add, "ellgpu_point_add");
  SYM(sig_from_der, "ellgpu_sig_from_der");
  SYM(sig_to_der, "ellgpu_sig_to_der");
  SYM(verify_wire, "ellgpu_ecdsa_verify_wire");
  L.h = h;
  napi_value t; CHECK(env, napi_get_boolean(env, 1, &t));
  return t;
}

/* Environment teardown (the script has ended, node is about to exit): finalizers still run then
 * (node >= 12.17), in no useful order with respect to the HIP runtime's own exit handlers and its
 * helper threads.  A context that is still alive at that point is not torn down piece by piece --
 * the process ends, the driver reclaims its device memory -- and the result-buffer pool stops
 * telling V8 about memory V8 no longer tracks.  (One of ~ 40 fuzz processes on a loaded GPU box
 * ended with SIGSEGV AFTER its last line of output; nothing but teardown was left to run.)
 * g_closing is set by an environment cleanup hook, which node runs before the finalizers. */
static int g_closing = 0;
static void on_env_cleanup(void* arg) { (void)arg; g_closing = 1; }
/* Contexts are PINNED: the addon holds a strong reference to every context external until
 * destroyContext(ctx) is called (Engine#close in index.js).  Left to the garbage collector, the
 * external of a script that has just run off its end is garbage a moment before node frees the
 * environment; when a collection falls into that moment, the reference's first-pass weak callback
 * has run and its second pass is a pending task while napi_env's teardown finalizes and deletes
 * the same reference -- and node 12.22 then runs the pending second pass on freed memory (caught
 * under rocgdb: FreeEnvironment -> RunCleanup -> CleanupHandles -> uv_run ->
 * InvokeSecondPassPhantomCallbacks -> N-API's reference helpers; fixed in node >= 14.17).  A strong
 * reference has no weak callback.  A context owns gigabytes of device memory anyway: its lifetime
 * is explicit, like that of the stream it wraps. */
#define MAX_CTX 64
static struct { ellgpu_ctx* c; napi_ref pin; } g_ctx[MAX_CTX];
static int g_jobs_in_flight = 0;
static int ctx_slot(ellgpu_ctx* c) {
  for (int i = 0; i < MAX_CTX; i++) if (g_ctx[i].c == c) return i;
  return -1;
}
static void ctx_finalize(napi_env env, void* data, void* hint) {
  (void)env; (void)hint;
  if (getenv("ELLGPU_NAPI_TRACE")) fprintf(stderr, "ellgpu_napi: ctx_finalize closing=%d\n", g_closing);
  /* (reached for the external of a context that destroyContext has released -- only then is it
   * unpinned and collectable -- and for every external at environment teardown: nothing is left to do) */
  (void)data;
}
static napi_value fn_create(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 1; napi_value argv[1]; int32_t dev = 0;
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = NULL;
  bool is_arr = false;
  if (argc >= 1) napi_is_array(env, argv[0], &is_arr);
  if (is_arr) {
    /* createContext([d0, d1, ...]): a device group (ellgpu_group_create) -- the batch entry
     * points shard over the devices, every other call runs on the first one */
    uint32_t nd = 0; int devs[64];
    CHECK(env, napi_get_array_length(env, argv[0], &nd));
    if (nd < 1 || nd > 64) THROW(env, "createContext([devices]): 1..64 devices");
    for (uint32_t i = 0; i < nd; i++) {
      napi_value e; int32_t d = 0;
      CHECK(env, napi_get_element(env, argv[0], i, &e));
      if (napi_get_value_int32(env, e, &d) != napi_ok) THROW(env, "createContext([devices]): integers expected");
      devs[i] = d;
    }
    if (L.group_create(devs, (int)nd, &c) != 0) return lib_error(env);
  } else {
    if (argc >= 1) napi_get_value_int32(env, argv[0], &dev);
    if (L.ctx_create(dev, &c) != 0) return lib_error(env);
  }
  int slot = ctx_slot(NULL);
  if (slot < 0) { L.ctx_destroy(c); THROW(env, "ellgpu: too many live contexts (destroyContext / Engine#close releases one)"); }
  napi_value ext;
  if (napi_create_external(env, c, ctx_finalize, NULL, &ext) != napi_ok ||
      napi_create_reference(env, ext, 1, &g_ctx[slot].pin) != napi_ok) { L.ctx_destroy(c); THROW(env, "ellgpu: could not wrap the context"); }
  g_ctx[slot].c = c;
  return ext;
}
static ellgpu_ctx* get_ctx(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_error(env, NULL, "ellgpu: bad context"); return NULL; }
  if (ctx_slot((ellgpu_ctx*)p) < 0) { napi_throw_error(env, NULL, "ellgpu: the context has been destroyed"); return NULL; }
  return (ellgpu_ctx*)p;
}
static int get_buf(napi_env env, napi_value v, const uint8_t** p, size_t* len, int allow_null) {
  napi_valuetype t; napi_typeof(env, v, &t);
  if (allow_null && (t == napi_null || t == napi_undefined)) { *p = NULL; *len = 0; return 1; }
  bool isbuf = 0; napi_is_buffer(env, v, &isbuf);
  if (!isbuf) { napi_throw_error(env, NULL, "ellgpu: Buffer expected"); return 0; }
  void* d; if (napi_get_buffer_info(env, v, &d, len) != napi_ok) { napi_throw_error(env, NULL, "ellgpu: bad Buffer"); return 0; }
  *p = (const uint8_t*)d; return 1;
}
/* ---- result buffers -------------------------------------------------------------------
 * A fresh n x 64-byte Buffer costs more to fault in (first touch, ~6 ms per 64 MB, whoever
 * touches it) than the GPU needs to fill it.  Large result Buffers are therefore external
 * Buffers over blocks that return to a free list when V8 collects them: after the first call of
 * a given size the memory has been touched already.  The blocks are 2 MiB-aligned and advised
 * as huge pages (the copy engine moves device data into huge-page-backed memory ~3x faster
 * than into 4 KiB pages here; numpy does the same for its large arrays).  Callers that keep
 * their own result Buffers pass them in (mulFixed / mulVar / mulAdd2: trailing xy, inf
 * arguments) and skip all of this.  Everything here runs on the JS thread. */
typedef struct pool_blk { void* p; size_t cap; struct pool_blk* next; } pool_blk;
static pool_blk* g_pool = NULL;
static size_t g_pool_bytes = 0;
#define POOL_MIN_BYTES ((size_t)256 * 1024)
#define POOL_MAX_BYTES ((size_t)1 << 31)
static void pool_release(napi_env env, void* data, void* hint) {
  (void)data;
  pool_blk* b = (pool_blk*)hint;
  int64_t adj;
  if (g_closing) { free(b->p); free(b); return; }
  napi_adjust_external_memory(env, -(int64_t)b->cap, &adj);
  if (g_pool_bytes + b->cap > POOL_MAX_BYTES) { free(b->p); free(b); return; }
  b->next = g_pool; g_pool = b; g_pool_bytes += b->cap;
}
static napi_status result_buffer(napi_env env, size_t size, void** data, napi_value* out) {
  if (size < POOL_MIN_BYTES) return napi_create_buffer(env, size, data, out);
  if (getenv("ELLGPU_NAPI_TRACE")) fprintf(stderr, "ellgpu_napi: pooled result buffer of %zu bytes\n", size);
  pool_blk **pp = &g_pool, **best = NULL;
  for (; *pp; pp = &(*pp)->next)                         /* best fit, at most 2x the request */
    if ((*pp)->cap >= size && (*pp)->cap <= 2 * size && (!best || (*pp)->cap < (*best)->cap)) best = pp;
  pool_blk* b;
  if (best) {
    b = *best; *best = b->next; g_pool_bytes -= b->cap;
  } else {
    b = (pool_blk*)malloc(sizeof *b);
    if (!b) return napi_generic_failure;
    b->cap = (size + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (posix_memalign(&b->p, (size_t)2 << 20, b->cap) != 0) { free(b); return napi_generic_failure; }
    madvise(b->p, b->cap, MADV_HUGEPAGE);                /* as numpy does for large arrays */
    memset(b->p, 0, b->cap);                             /* first touch here, once per block */
  }
  b->next = NULL;
  *data = b->p;
  napi_status st = napi_create_external_buffer(env, size, b->p, pool_release, b, out);
  if (st != napi_ok) { free(b->p); free(b); return st; }
  int64_t adj;                                           /* let V8 see the memory behind the Buffer */
  napi_adjust_external_memory(env, (int64_t)b->cap, &adj);
  return st;
}

static napi_value mk_result(napi_env env, const char* k1, napi_value a, const char* k2, napi_value b) {
  napi_value o; CHECK(env, napi_create_object(env, &o));
  CHECK(env, napi_set_named_property(env, o, k1, a));
  CHECK(env, napi_set_named_property(env, o, k2, b));
  return o;
}
static napi_value fn_curve_id(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 1; napi_value argv[1]; char name[64]; size_t len;
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &len) != napi_ok) THROW(env, "curveId(name)");
  napi_value r; CHECK(env, napi_create_int32(env, L.curve_id(name), &r)); return r;
}
static napi_value int_fn(napi_env env, napi_callback_info info, int which) {
  if (!need_lib(env)) return NULL;
  size_t argc = 1; napi_value argv[1]; int32_t id = 0;
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc >= 1) napi_get_value_int32(env, argv[0], &id);
  int v = which == 0 ? L.field_bytes(id) : which == 1 ? L.order_bytes(id) : L.device_count();
  napi_value r; CHECK(env, napi_create_int32(env, v, &r)); return r;
}
static napi_value fn_field_bytes(napi_env e, napi_callback_info i) { return int_fn(e, i, 0); }
static napi_value fn_order_bytes(napi_env e, napi_callback_info i) { return int_fn(e, i, 1); }
static napi_value fn_device_count(napi_env e, napi_callback_info i) { return int_fn(e, i, 2); }

static napi_value fn_group_size(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 1; napi_value argv[1];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  napi_value v; CHECK(env, napi_create_int32(env, L.group_size(c), &v));
  return v;
}
/* defineShort(ctx, p, a, b) / defineEdwards(ctx, p, a, d) -> curve id: 32-byte big-endian Buffers
 * (ellgpu_curve_define_short / ellgpu_curve_define_edwards) */
static napi_value define_common(napi_env env, napi_callback_info info, int edwards) {
  if (!need_lib(env)) return NULL;
  size_t argc = 4; napi_value argv[4];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 4) THROW(env, "defineShort(ctx, p, a, b)");
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  const uint8_t* b[3]; size_t l[3];
  for (int i = 0; i < 3; i++) {
    if (!get_buf(env, argv[1 + i], &b[i], &l[i], 0)) return NULL;
    if (l[i] != 32) THROW(env, "defineShort: p, a, b are 32-byte big-endian Buffers");
  }
  int id = -1;
  if ((edwards ? L.define_edwards : L.define_short)(c, b[0], b[1], b[2], &id) != 0) THROW(env, L.last_error());
  napi_value v; CHECK(env, napi_create_int32(env, id, &v));
  return v;
}
static napi_value fn_define_short(napi_env e, napi_callback_info i) { return define_common(e, i, 0); }
static napi_value fn_define_edwards(napi_env e, napi_callback_info i) { return define_common(e, i, 1); }
/* destroyContext(ctx): releases the context's device memory and streams (ellgpu_ctx_destroy) and the
 * addon's pin on the external; any later call with that external throws.  Refused while Promise-form
 * batches are in flight (a worker thread is inside the context). */
static napi_value fn_destroy(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 1; napi_value argv[1];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* p = NULL;
  napi_value u; napi_get_undefined(env, &u);
  if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) THROW(env, "destroyContext(ctx)");
  int i = ctx_slot((ellgpu_ctx*)p);
  if (i < 0) return u;                                  /* destroyed already */
  if (g_jobs_in_flight > 0) THROW(env, "ellgpu: batches are in flight on a worker thread; destroy the context when their Promises have settled");
  L.ctx_destroy(g_ctx[i].c);
  napi_delete_reference(env, g_ctx[i].pin);
  g_ctx[i].c = NULL; g_ctx[i].pin = NULL;
  return u;
}

/* defer(ctx) / collect(ctx): the split form of a few-item call (ellgpu_ctx_defer / _collect) */
static napi_value defer_common(napi_env env, napi_callback_info info, int collect) {
  if (!need_lib(env)) return NULL;
  size_t argc = 1; napi_value argv[1];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  if ((collect ? L.ctx_collect : L.ctx_defer)(c) != 0) return lib_error(env);
  napi_value u; napi_get_undefined(env, &u); return u;
}
static napi_value fn_defer(napi_env e, napi_callback_info i) { return defer_common(e, i, 0); }
static napi_value fn_collect(napi_env e, napi_callback_info i) { return defer_common(e, i, 1); }

/* combBits(ctx, curve) -> window width of the curve's fixed-base table on this context (0: not
 * built yet; narrower than the default when the device could not hold it: ellgpu_ctx_comb_bits) */
static napi_value fn_comb_bits(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 2; napi_value argv[2]; int32_t curve = 0;
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  if (argc < 2 || napi_get_value_int32(env, argv[1], &curve) != napi_ok) THROW(env, "combBits(ctx, curve)");
  int v = L.ctx_comb_bits(c, curve);
  if (v < 0) return lib_error(env);
  napi_value r; CHECK(env, napi_create_int32(env, v, &r)); return r;
}

/* mulFixed / mulVar / mulAdd2 share the output shape */
static napi_value mul_common(napi_env env, napi_callback_info info, int kind) {
  if (!need_lib(env)) return NULL;
  size_t argc = 8; napi_value argv[8];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve; if (napi_get_value_int32(env, argv[1], &curve) != napi_ok) THROW(env, "curve id expected");
  int B = L.field_bytes(curve); if (B <= 0) THROW(env, "unknown curve id");
  const uint8_t *k1 = 0, *p1 = 0, *k2 = 0, *p2 = 0; size_t l1 = 0, lp1 = 0, l2 = 0, lp2 = 0;
  if (!get_buf(env, argv[2], &k1, &l1, 0)) return NULL;
  if (l1 % (size_t)B) THROW(env, "scalar buffer length is not a multiple of the field width");
  size_t n = l1 / (size_t)B;
  if (kind == 1) { if (!get_buf(env, argv[3], &p1, &lp1, 0)) return NULL; if (lp1 != n * 2 * (size_t)B) THROW(env, "point buffer length mismatch"); }
  if (kind == 2) {
    if (!get_buf(env, argv[3], &p1, &lp1, 1)) return NULL;
    if (!get_buf(env, argv[4], &k2, &l2, 0)) return NULL;
    if (!get_buf(env, argv[5], &p2, &lp2, 0)) return NULL;
    if ((p1 && lp1 != n * 2 * (size_t)B) || l2 != l1 || lp2 != n * 2 * (size_t)B) THROW(env, "buffer length mismatch");
  }
  napi_value bxy, binf; void *dxy, *dinf;
  /* optional trailing (xy, inf) result Buffers supplied by the caller */
  size_t oi = kind == 0 ? 3 : kind == 1 ? 4 : 6;
  bool own = false;
  if (argc >= oi + 2) {
    bool is0 = false, is1 = false;
    napi_is_buffer(env, argv[oi], &is0); napi_is_buffer(env, argv[oi + 1], &is1);
    own = is0 && is1;
  }
  if (own) {
    size_t lx, li;
    CHECK(env, napi_get_buffer_info(env, argv[oi], &dxy, &lx));
    CHECK(env, napi_get_buffer_info(env, argv[oi + 1], &dinf, &li));
    if (lx != n * 2 * (size_t)B || li != n) THROW(env, "result buffer length mismatch");
    bxy = argv[oi]; binf = argv[oi + 1];
  } else {
    CHECK(env, result_buffer(env, n * 2 * (size_t)B, &dxy, &bxy));
    CHECK(env, result_buffer(env, n, &dinf, &binf));
  }
  int rc = kind == 0 ? L.mul_fixed(c, curve, n, k1, (uint8_t*)dxy, (uint8_t*)dinf)
         : kind == 1 ? L.mul_var(c, curve, n, k1, p1, (uint8_t*)dxy, (uint8_t*)dinf)
                     : L.mul_add2(c, curve, n, k1, p1, k2, p2, (uint8_t*)dxy, (uint8_t*)dinf);
  if (rc != 0) return lib_error(env);
  return mk_result(env, "xy", bxy, "inf", binf);
}
static napi_value fn_mul_fixed(napi_env e, napi_callback_info i) { return mul_common(e, i, 0); }
static napi_value fn_mul_var(napi_env e, napi_callback_info i) { return mul_common(e, i, 1); }
static napi_value fn_mul_add2(napi_env e, napi_callback_info i) { return mul_common(e, i, 2); }

static napi_value fn_verify(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 8; napi_value argv[8];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve, hl, mb;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_int32(env, argv[3], &hl) != napi_ok ||
      napi_get_value_int32(env, argv[4], &mb) != napi_ok) THROW(env, "ecdsaVerify(ctx, curve, hash, hashLen, msgBits, r, s, pub)");
  int B = L.field_bytes(curve), NB = L.order_bytes(curve);
  if (B <= 0 || hl <= 0) THROW(env, "bad curve / hashLen");
  const uint8_t *h, *r, *s, *q; size_t lh, lr, ls, lq;
  if (!get_buf(env, argv[2], &h, &lh, 0) || !get_buf(env, argv[5], &r, &lr, 0) ||
      !get_buf(env, argv[6], &s, &ls, 0) || !get_buf(env, argv[7], &q, &lq, 0)) return NULL;
  if (lh % (size_t)hl) THROW(env, "hash buffer length is not a multiple of hashLen");
  size_t n = lh / (size_t)hl;
  if (lr != n * (size_t)NB || ls != n * (size_t)NB || lq != n * 2 * (size_t)B) THROW(env, "buffer length mismatch");
  napi_value bok, bst; void *dok, *dst;
  CHECK(env, result_buffer(env, n, &dok, &bok));
  CHECK(env, result_buffer(env, n, &dst, &bst));
  if (L.ecdsa_verify(c, curve, n, h, hl, mb, r, s, q, (uint8_t*)dok, (uint8_t*)dst) != 0) return lib_error(env);
  return mk_result(env, "ok", bok, "status", bst);
}
static napi_value fn_x25519(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 3; napi_value argv[3];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  const uint8_t *k, *x; size_t lk, lx;
  if (!get_buf(env, argv[1], &k, &lk, 0) || !get_buf(env, argv[2], &x, &lx, 0)) return NULL;
  if (lk % 32 || lx != lk) THROW(env, "buffer length mismatch");
  size_t n = lk / 32;
  napi_value bx, binf; void *dx, *dinf;
  CHECK(env, result_buffer(env, n * 32, &dx, &bx));
  CHECK(env, result_buffer(env, n, &dinf, &binf));
  if (L.x25519(c, n, k, x, (uint8_t*)dx, (uint8_t*)dinf) != 0) return lib_error(env);
  return mk_result(env, "x", bx, "inf", binf);
}

/* x25519Derive(ctx, k, x) -> {x, status}: KeyPair#derive on curve25519 (ellgpu_x25519_derive) */
static napi_value fn_x25519_derive(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 3; napi_value argv[3];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  const uint8_t *k, *x; size_t lk, lx;
  if (!get_buf(env, argv[1], &k, &lk, 0) || !get_buf(env, argv[2], &x, &lx, 0)) return NULL;
  if (lk % 32 || lx != lk) THROW(env, "buffer length mismatch");
  size_t n = lk / 32;
  napi_value bx, bst; void *dx, *dst;
  CHECK(env, result_buffer(env, n * 32, &dx, &bx));
  CHECK(env, result_buffer(env, n, &dst, &bst));
  if (L.x25519_derive(c, n, k, x, (uint8_t*)dx, (uint8_t*)dst) != 0) return lib_error(env);
  return mk_result(env, "x", bx, "status", bst);
}

static napi_value fn_decompress(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 4; napi_value argv[4];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve; if (napi_get_value_int32(env, argv[1], &curve) != napi_ok) THROW(env, "curve id expected");
  int B = L.field_bytes(curve); if (B <= 0) THROW(env, "unknown curve id");
  const uint8_t *v, *odd; size_t lv, lo;
  if (!get_buf(env, argv[2], &v, &lv, 0) || !get_buf(env, argv[3], &odd, &lo, 0)) return NULL;
  if (lv % (size_t)B || lo != lv / (size_t)B) THROW(env, "buffer length mismatch");
  size_t n = lo;
  napi_value bxy, bok; void *dxy, *dok;
  CHECK(env, result_buffer(env, n * 2 * (size_t)B, &dxy, &bxy));
  CHECK(env, result_buffer(env, n, &dok, &bok));
  if (L.decompress(c, curve, n, v, odd, (uint8_t*)dxy, (uint8_t*)dok) != 0) return lib_error(env);
  return mk_result(env, "xy", bxy, "ok", bok);
}

static napi_value fn_sign(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 8; napi_value argv[8];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve, hl, mb; bool canon = 0;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_int32(env, argv[3], &hl) != napi_ok ||
      napi_get_value_int32(env, argv[4], &mb) != napi_ok) THROW(env, "ecdsaSign(ctx, curve, hash, hashLen, msgBits, priv, nonces, canonical)");
  napi_get_value_bool(env, argv[7], &canon);
  int NB = L.order_bytes(curve);
  if (NB <= 0 || hl <= 0) THROW(env, "bad curve / hashLen");
  const uint8_t *h, *d, *k; size_t lh, ld, lk;
  if (!get_buf(env, argv[2], &h, &lh, 0) || !get_buf(env, argv[5], &d, &ld, 0) || !get_buf(env, argv[6], &k, &lk, 0)) return NULL;
  if (lh % (size_t)hl) THROW(env, "hash buffer length is not a multiple of hashLen");
  size_t n = lh / (size_t)hl;
  if (ld != n * (size_t)NB || lk != ld) THROW(env, "buffer length mismatch");
  napi_value br, bs, brec, bok, o; void *dr, *dsg, *drec, *dok;
  CHECK(env, result_buffer(env, n * (size_t)NB, &dr, &br));
  CHECK(env, result_buffer(env, n * (size_t)NB, &dsg, &bs));
  CHECK(env, result_buffer(env, n, &drec, &brec));
  CHECK(env, result_buffer(env, n, &dok, &bok));
  if (L.ecdsa_sign(c, curve, n, h, hl, mb, d, k, canon ? 1 : 0, (uint8_t*)dr, (uint8_t*)dsg, (uint8_t*)drec, (uint8_t*)dok) != 0)
    return lib_error(env);
  CHECK(env, napi_create_object(env, &o));
  CHECK(env, napi_set_named_property(env, o, "r", br));
  CHECK(env, napi_set_named_property(env, o, "s", bs));
  CHECK(env, napi_set_named_property(env, o, "recid", brec));
  CHECK(env, napi_set_named_property(env, o, "ok", bok));
  return o;
}

/* ecdsaSignDet(ctx, curve, hash, hashLen, msgBits, priv, canonical) -> {r, s, recid, ok} */
static napi_value fn_sign_det(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 7; napi_value argv[7];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve, hl, mb; bool canon = 0;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_int32(env, argv[3], &hl) != napi_ok ||
      napi_get_value_int32(env, argv[4], &mb) != napi_ok) THROW(env, "ecdsaSignDet(ctx, curve, hash, hashLen, msgBits, priv, canonical)");
  napi_get_value_bool(env, argv[6], &canon);
  int NB = L.order_bytes(curve);
  if (NB <= 0 || hl <= 0) THROW(env, "bad curve / hashLen");
  const uint8_t *h, *d; size_t lh, ld;
  if (!get_buf(env, argv[2], &h, &lh, 0) || !get_buf(env, argv[5], &d, &ld, 0)) return NULL;
  if (lh % (size_t)hl) THROW(env, "hash buffer length is not a multiple of hashLen");
  size_t n = lh / (size_t)hl;
  if (ld != n * (size_t)NB) THROW(env, "buffer length mismatch");
  napi_value br, bs, brec, bok, o; void *dr, *dsg, *drec, *dok;
  CHECK(env, result_buffer(env, n * (size_t)NB, &dr, &br));
  CHECK(env, result_buffer(env, n * (size_t)NB, &dsg, &bs));
  CHECK(env, result_buffer(env, n, &drec, &brec));
  CHECK(env, result_buffer(env, n, &dok, &bok));
  if (L.ecdsa_sign_det(c, curve, n, h, hl, mb, d, canon ? 1 : 0, (uint8_t*)dr, (uint8_t*)dsg, (uint8_t*)drec, (uint8_t*)dok) != 0)
    return lib_error(env);
  CHECK(env, napi_create_object(env, &o));
  CHECK(env, napi_set_named_property(env, o, "r", br));
  CHECK(env, napi_set_named_property(env, o, "s", bs));
  CHECK(env, napi_set_named_property(env, o, "recid", brec));
  CHECK(env, napi_set_named_property(env, o, "ok", bok));
  return o;
}

/* ecdsaRecover(ctx, curve, hash, hashLen, r, s, recid) -> {xy: Buffer(n*2B), status: Buffer(n)} */
static napi_value fn_recover(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 7; napi_value argv[7];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve, hl;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_int32(env, argv[3], &hl) != napi_ok)
    THROW(env, "ecdsaRecover(ctx, curve, hash, hashLen, r, s, recid)");
  int B = L.field_bytes(curve), NB = L.order_bytes(curve);
  if (B <= 0 || NB <= 0 || hl <= 0) THROW(env, "bad curve / hashLen");
  const uint8_t *h, *r, *sg, *j; size_t lh, lr, ls, lj;
  if (!get_buf(env, argv[2], &h, &lh, 0) || !get_buf(env, argv[4], &r, &lr, 0) ||
      !get_buf(env, argv[5], &sg, &ls, 0) || !get_buf(env, argv[6], &j, &lj, 0)) return NULL;
  if (lh % (size_t)hl) THROW(env, "hash buffer length is not a multiple of hashLen");
  size_t n = lh / (size_t)hl;
  if (lr != n * (size_t)NB || ls != lr || lj != n) THROW(env, "buffer length mismatch");
  napi_value bxy, bst; void *dxy, *dst;
  CHECK(env, result_buffer(env, n * 2 * (size_t)B, &dxy, &bxy));
  CHECK(env, result_buffer(env, n, &dst, &bst));
  if (L.ecdsa_recover(c, curve, n, h, hl, r, sg, j, (uint8_t*)dxy, (uint8_t*)dst) != 0) return lib_error(env);
  return mk_result(env, "xy", bxy, "status", bst);
}

static napi_value fn_decode_points(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 4; napi_value argv[4];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve, el;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_int32(env, argv[3], &el) != napi_ok)
    THROW(env, "decodePoints(ctx, curve, enc, encLen)");
  int B = L.field_bytes(curve); if (B <= 0 || el <= 0) THROW(env, "bad curve / encLen");
  const uint8_t* enc; size_t le;
  if (!get_buf(env, argv[2], &enc, &le, 0)) return NULL;
  if (le % (size_t)el) THROW(env, "enc buffer length is not a multiple of encLen");
  size_t n = le / (size_t)el;
  napi_value bxy, bst; void *dxy, *dst;
  CHECK(env, result_buffer(env, n * 2 * (size_t)B, &dxy, &bxy));
  CHECK(env, result_buffer(env, n, &dst, &bst));
  if (L.decode_points(c, curve, n, enc, (size_t)el, (uint8_t*)dxy, (uint8_t*)dst) != 0) return lib_error(env);
  return mk_result(env, "xy", bxy, "status", bst);
}

static napi_value fn_encode_points(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 4; napi_value argv[4];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve; bool compact = 0;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_bool(env, argv[3], &compact) != napi_ok)
    THROW(env, "encodePoints(ctx, curve, xy, compact)");
  int B = L.field_bytes(curve); if (B <= 0) THROW(env, "unknown curve id");
  const uint8_t* xy; size_t lx;
  if (!get_buf(env, argv[2], &xy, &lx, 0)) return NULL;
  if (lx % (2 * (size_t)B)) THROW(env, "xy buffer length is not a multiple of 2 * fieldBytes");
  size_t n = lx / (2 * (size_t)B);
  /* the library's rule for the encoded length: SEC1 for the short curves, 32 bytes for ed25519 */
  size_t el = compact ? 1 + (size_t)B : 1 + 2 * (size_t)B;
  if (L.curve_id("ed25519") == curve) el = 32;
  napi_value be; void* de;
  CHECK(env, result_buffer(env, n * el, &de, &be));
  if (L.encode_points(c, curve, n, xy, compact ? 1 : 0, (uint8_t*)de) != 0) return lib_error(env);
  return be;
}

static napi_value fn_validate(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 5; napi_value argv[5];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve; bool order = 1;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_bool(env, argv[4], &order) != napi_ok)
    THROW(env, "validate(ctx, curve, xy, inf|null, checkOrder)");
  int B = L.field_bytes(curve); if (B <= 0) THROW(env, "unknown curve id");
  const uint8_t *xy, *inf; size_t lx, li;
  if (!get_buf(env, argv[2], &xy, &lx, 0) || !get_buf(env, argv[3], &inf, &li, 1)) return NULL;
  if (lx % (2 * (size_t)B)) THROW(env, "xy buffer length is not a multiple of 2 * fieldBytes");
  size_t n = lx / (2 * (size_t)B);
  if (inf && li != n) THROW(env, "buffer length mismatch");
  napi_value bst; void* dst;
  CHECK(env, result_buffer(env, n, &dst, &bst));
  if (L.validate(c, curve, n, xy, inf, order ? 1 : 0, (uint8_t*)dst) != 0) return lib_error(env);
  return bst;
}

static napi_value fn_point_add(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 6; napi_value argv[6];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok) THROW(env, "pointAdd(ctx, curve, xy1, inf1|null, xy2, inf2|null)");
  int B = L.field_bytes(curve); if (B <= 0) THROW(env, "unknown curve id");
  const uint8_t *p1, *i1, *p2, *i2; size_t l1, li1, l2, li2;
  if (!get_buf(env, argv[2], &p1, &l1, 0) || !get_buf(env, argv[3], &i1, &li1, 1) ||
      !get_buf(env, argv[4], &p2, &l2, 0) || !get_buf(env, argv[5], &i2, &li2, 1)) return NULL;
  if (l1 % (2 * (size_t)B) || l2 != l1) THROW(env, "buffer length mismatch");
  size_t n = l1 / (2 * (size_t)B);
  if ((i1 && li1 != n) || (i2 && li2 != n)) THROW(env, "buffer length mismatch");
  napi_value bxy, binf; void *dxy, *dinf;
  CHECK(env, result_buffer(env, n * 2 * (size_t)B, &dxy, &bxy));
  CHECK(env, result_buffer(env, n, &dinf, &binf));
  if (L.point_add(c, curve, n, p1, i1, p2, i2, (uint8_t*)dxy, (uint8_t*)dinf) != 0) return lib_error(env);
  return mk_result(env, "xy", bxy, "inf", binf);
}

static napi_value fn_sig_from_der(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 5; napi_value argv[5];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve, stride;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_int32(env, argv[3], &stride) != napi_ok)
    THROW(env, "sigFromDer(ctx, curve, der, stride, lens)");
  int NB = L.order_bytes(curve); if (NB <= 0 || stride <= 0) THROW(env, "bad curve / stride");
  const uint8_t *der, *lens; size_t ld, ll;
  if (!get_buf(env, argv[2], &der, &ld, 0) || !get_buf(env, argv[4], &lens, &ll, 0)) return NULL;
  if (ll % 4 || ((uintptr_t)lens & 3)) THROW(env, "lens must be an aligned Buffer of uint32");
  size_t n = ll / 4;
  if (ld != n * (size_t)stride) THROW(env, "buffer length mismatch");
  napi_value br, bs, bst, o; void *dr, *ds, *dst;
  CHECK(env, result_buffer(env, n * (size_t)NB, &dr, &br));
  CHECK(env, result_buffer(env, n * (size_t)NB, &ds, &bs));
  CHECK(env, result_buffer(env, n, &dst, &bst));
  if (L.sig_from_der(c, curve, n, der, (size_t)stride, (const uint32_t*)lens, (uint8_t*)dr, (uint8_t*)ds,
                     (uint8_t*)dst) != 0) return lib_error(env);
  o = mk_result(env, "r", br, "s", bs); if (!o) return NULL;
  CHECK(env, napi_set_named_property(env, o, "status", bst));
  return o;
}

static napi_value fn_sig_to_der(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 4; napi_value argv[4];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok) THROW(env, "sigToDer(ctx, curve, r, s)");
  int NB = L.order_bytes(curve); if (NB <= 0) THROW(env, "unknown curve id");
  const uint8_t *r, *s; size_t lr, ls;
  if (!get_buf(env, argv[2], &r, &lr, 0) || !get_buf(env, argv[3], &s, &ls, 0)) return NULL;
  if (lr % (size_t)NB || ls != lr) THROW(env, "buffer length mismatch");
  size_t n = lr / (size_t)NB, stride = 2 * (size_t)NB + 9;
  napi_value bd, bl; void *dd, *dl;
  CHECK(env, result_buffer(env, n * stride, &dd, &bd));
  CHECK(env, result_buffer(env, n * 4, &dl, &bl));
  if (L.sig_to_der(c, curve, n, r, s, (uint8_t*)dd, stride, (uint32_t*)dl) != 0) return lib_error(env);
  return mk_result(env, "der", bd, "lens", bl);
}

static napi_value fn_verify_wire(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 10; napi_value argv[10];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  int32_t curve, hl, mb, stride, kl;
  if (napi_get_value_int32(env, argv[1], &curve) != napi_ok || napi_get_value_int32(env, argv[3], &hl) != napi_ok ||
      napi_get_value_int32(env, argv[4], &mb) != napi_ok || napi_get_value_int32(env, argv[6], &stride) != napi_ok ||
      napi_get_value_int32(env, argv[9], &kl) != napi_ok)
    THROW(env, "ecdsaVerifyWire(ctx, curve, hash, hashLen, msgBits, der, stride, lens, keys, keyLen)");
  if (hl <= 0 || stride <= 0 || kl <= 0) THROW(env, "bad hashLen / stride / keyLen");
  const uint8_t *h, *der, *lens, *keys; size_t lh, ld, ll, lk;
  if (!get_buf(env, argv[2], &h, &lh, 0) || !get_buf(env, argv[5], &der, &ld, 0) ||
      !get_buf(env, argv[7], &lens, &ll, 0) || !get_buf(env, argv[8], &keys, &lk, 0)) return NULL;
  if (lh % (size_t)hl) THROW(env, "hash buffer length is not a multiple of hashLen");
  size_t n = lh / (size_t)hl;
  if (ll != n * 4 || ((uintptr_t)lens & 3)) THROW(env, "lens must be an aligned Buffer of n uint32");
  if (ld != n * (size_t)stride || lk != n * (size_t)kl) THROW(env, "buffer length mismatch");
  napi_value bok, berr; void *dok, *derr;
  CHECK(env, result_buffer(env, n, &dok, &bok));
  CHECK(env, result_buffer(env, n, &derr, &berr));
  if (L.verify_wire(c, curve, n, h, hl, mb, der, (size_t)stride, (const uint32_t*)lens, keys, (size_t)kl,
                    (uint8_t*)dok, (uint8_t*)derr) != 0) return lib_error(env);
  return mk_result(env, "ok", bok, "err", berr);
}

static napi_value fn_eddsa_verify(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 6; napi_value argv[6];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  const uint8_t *m, *off, *sg, *pk; size_t lm, loff, lsg, lpk; int32_t mlen = 0;
  if (!get_buf(env, argv[1], &m, &lm, 1) || !get_buf(env, argv[2], &off, &loff, 1) ||
      !get_buf(env, argv[4], &sg, &lsg, 0) || !get_buf(env, argv[5], &pk, &lpk, 0)) return NULL;
  napi_get_value_int32(env, argv[3], &mlen);
  if (lsg % 64 || lpk != lsg / 2) THROW(env, "buffer length mismatch");
  size_t n = lsg / 64;
  if (off) {
    if (loff != (n + 1) * 8 || ((uintptr_t)off & 7)) THROW(env, "offsets must be an aligned Buffer of n+1 uint64");
    if (((const uint64_t*)off)[n] > lm) THROW(env, "offsets exceed the message buffer");
    for (size_t i = 0; i < n; i++)        /* the kernels take len = off[i+1] - off[i]: must not wrap */
      if (((const uint64_t*)off)[i] > ((const uint64_t*)off)[i + 1]) THROW(env, "offsets must be non-decreasing");
  } else if ((size_t)mlen * n > lm) THROW(env, "message buffer too short");
  napi_value bok, berr; void *dok, *derr;
  CHECK(env, result_buffer(env, n, &dok, &bok));
  CHECK(env, result_buffer(env, n, &derr, &berr));
  if (L.eddsa_verify(c, n, m, (const uint64_t*)off, (size_t)mlen, sg, pk, (uint8_t*)dok, (uint8_t*)derr) != 0)
    return lib_error(env);
  return mk_result(env, "ok", bok, "err", berr);
}

/* eddsaSign(ctx, msgs, offsets|null, msgLen, secrets) -> {sig: Buffer(n*64), pub: Buffer(n*32)} */
static napi_value fn_eddsa_sign(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 5; napi_value argv[5];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  ellgpu_ctx* c = get_ctx(env, argv[0]); if (!c) return NULL;
  const uint8_t *m, *off, *sec; size_t lm, loff, lsec; int32_t mlen = 0;
  if (!get_buf(env, argv[1], &m, &lm, 1) || !get_buf(env, argv[2], &off, &loff, 1) ||
      !get_buf(env, argv[4], &sec, &lsec, 0)) return NULL;
  napi_get_value_int32(env, argv[3], &mlen);
  if (lsec % 32) THROW(env, "secrets must be n x 32 bytes");
  size_t n = lsec / 32;
  if (off) {
    if (loff != (n + 1) * 8 || ((uintptr_t)off & 7)) THROW(env, "offsets must be an aligned Buffer of n+1 uint64");
    if (((const uint64_t*)off)[n] > lm) THROW(env, "offsets exceed the message buffer");
    for (size_t i = 0; i < n; i++)        /* the kernels take len = off[i+1] - off[i]: must not wrap */
      if (((const uint64_t*)off)[i] > ((const uint64_t*)off)[i + 1]) THROW(env, "offsets must be non-decreasing");
  } else if ((size_t)mlen * n > lm) THROW(env, "message buffer too short");
  napi_value bsig, bpub; void *dsig, *dpub;
  CHECK(env, result_buffer(env, n * 64, &dsig, &bsig));
  CHECK(env, result_buffer(env, n * 32, &dpub, &bpub));
  if (L.eddsa_sign(c, n, sec, m, (const uint64_t*)off, (size_t)mlen, (uint8_t*)dsig, (uint8_t*)dpub) != 0)
    return lib_error(env);
  return mk_result(env, "sig", bsig, "pub", bpub);
}

/* ---- asynchronous form: napi_async_work + Promise ---------------------------------- */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref refs[5];          /* four input Buffers + the context external */
  int nrefs;
  int op, curve, hash_len, msg_bits, B, NB;
  ellgpu_ctx* ctx;
  const uint8_t* in[4];
  size_t n;
  uint8_t* out0; size_t out0_len;      /* xy / ok / x */
  uint8_t* out1; size_t out1_len;      /* inf */
  uint8_t* out2; size_t out2_len;      /* ops 5..: third / fourth result */
  uint8_t* out3; size_t out3_len;
  int i0, i1;                          /* op 5: canonical; op 7: der stride, key length; op 8: encoding length */
  int rc;
  char err[512];
} async_job;

static void job_execute(napi_env env, void* data) {
  (void)env;
  async_job* j = (async_job*)data;
  switch (j->op) {
    case 0: j->rc = L.mul_fixed(j->ctx, j->curve, j->n, j->in[0], j->out0, j->out1); break;
    case 1: j->rc = L.mul_var(j->ctx, j->curve, j->n, j->in[0], j->in[1], j->out0, j->out1); break;
    case 2: j->rc = L.mul_add2(j->ctx, j->curve, j->n, j->in[0], j->in[1], j->in[2], j->in[3], j->out0, j->out1); break;
    case 3: j->rc = L.ecdsa_verify(j->ctx, j->curve, j->n, j->in[0], j->hash_len, j->msg_bits, j->in[1], j->in[2], j->in[3], j->out0, j->out1); break;
    case 4: j->rc = L.x25519(j->ctx, j->n, j->in[0], j->in[1], j->out0, j->out1); break;
    case 5: j->rc = L.ecdsa_sign_det(j->ctx, j->curve, j->n, j->in[0], j->hash_len, j->msg_bits, j->in[1], j->i0,
                                     j->out0, j->out1, j->out2, j->out3); break;
    case 6: j->rc = L.ecdsa_recover(j->ctx, j->curve, j->n, j->in[0], j->hash_len, j->in[1], j->in[2], j->in[3],
                                    j->out0, j->out1); break;
    case 7: j->rc = L.verify_wire(j->ctx, j->curve, j->n, j->in[0], j->hash_len, j->msg_bits, j->in[1], (size_t)j->i0,
                                  (const uint32_t*)j->in[2], j->in[3], (size_t)j->i1, j->out0, j->out1); break;
    default: j->rc = L.decode_points(j->ctx, j->curve, j->n, j->in[0], (size_t)j->i0, j->out0, j->out1); break;
  }
  if (j->rc != 0) {               /* last_error is thread-local: read it on this thread */
    const char* m = L.last_error();
    snprintf(j->err, sizeof j->err, "%s", m && *m ? m : "ellgpu error");
  }
}
static void free_cb(napi_env env, void* data, void* hint) { (void)env; (void)hint; free(data); }
static void job_complete(napi_env env, napi_status status, void* data) {
  async_job* j = (async_job*)data;
  napi_value result = NULL;
  if (status == napi_ok && j->rc == 0) {
    /* result property names per op, in output order */
    static const char* const names[9][4] = {
      {"xy", "inf", 0, 0}, {"xy", "inf", 0, 0}, {"xy", "inf", 0, 0}, {"ok", "status", 0, 0}, {"x", "inf", 0, 0},
      {"r", "s", "recid", "ok"}, {"xy", "status", 0, 0}, {"ok", "err", 0, 0}, {"xy", "status", 0, 0}};
    uint8_t** outs[4] = {&j->out0, &j->out1, &j->out2, &j->out3};
    size_t lens[4] = {j->out0_len, j->out1_len, j->out2_len, j->out3_len};
    napi_create_object(env, &result);
    for (int k = 0; k < 4 && names[j->op][k]; k++) {
      napi_value b;
      napi_create_external_buffer(env, lens[k], *outs[k], free_cb, NULL, &b);
      *outs[k] = NULL;
      napi_set_named_property(env, result, names[j->op][k], b);
    }
    napi_resolve_deferred(env, j->deferred, result);
  } else {
    napi_value msg, errv;
    napi_create_string_utf8(env, j->rc ? j->err : "ellgpu: async work cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &errv);
    napi_reject_deferred(env, j->deferred, errv);
  }
  for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
  g_jobs_in_flight--;
  napi_delete_async_work(env, j->work);
  free(j->out0); free(j->out1); free(j->out2); free(j->out3); free(j);
}
/* every early exit of fn_call_async: release the references taken so far (the context external
 * first of all -- a leaked reference keeps ctx_finalize / ellgpu_ctx_destroy from ever running) */
static void drop_job_refs(napi_env env, async_job* j) {
  for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
  j->nrefs = 0;
}
static napi_value fn_call_async(napi_env env, napi_callback_info info) {
  if (!need_lib(env)) return NULL;
  size_t argc = 11; napi_value argv[11];
  CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 9) THROW(env, "callAsync(op, ctx, curve, hashLen, msgBits, b0, b1, b2, b3[, i0, i1])");
  async_job* j = (async_job*)calloc(1, sizeof *j);
  int32_t op, curve, hl, mb;
  napi_get_value_int32(env, argv[0], &op);
  if (!j) THROW(env, "callAsync: out of memory");
  j->ctx = get_ctx(env, argv[1]);
  if (!j->ctx) { free(j); return NULL; }
  /* the worker thread uses the context after this call returns: hold its external so that the
   * GC finalizer (ctx_finalize -> ellgpu_ctx_destroy) cannot run under the job */
  napi_create_reference(env, argv[1], 1, &j->refs[j->nrefs++]);
  napi_get_value_int32(env, argv[2], &curve); napi_get_value_int32(env, argv[3], &hl); napi_get_value_int32(env, argv[4], &mb);
  j->op = op; j->curve = op == 4 ? 7 : curve; j->hash_len = hl; j->msg_bits = mb;
  j->B = L.field_bytes(j->curve); j->NB = L.order_bytes(j->curve);
  if (op < 0 || op > 8 || j->B <= 0) { drop_job_refs(env, j); free(j); THROW(env, "callAsync: bad op / curve"); }
  int32_t i0 = 0, i1 = 0;
  if (argc > 9) napi_get_value_int32(env, argv[9], &i0);
  if (argc > 10) napi_get_value_int32(env, argv[10], &i1);
  j->i0 = i0; j->i1 = i1;
  size_t len[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    if (!get_buf(env, argv[5 + i], &j->in[i], &len[i], 1)) { drop_job_refs(env, j); free(j); return NULL; }
    if (j->in[i]) napi_create_reference(env, argv[5 + i], 1, &j->refs[j->nrefs++]);   /* keep alive */
  }
  size_t B = (size_t)j->B, NB = (size_t)j->NB;
  int ok = 1;
  switch (op) {
    case 0: j->n = len[0] / B; ok = j->in[0] && len[0] % B == 0; break;
    case 1: j->n = len[0] / B; ok = j->in[0] && j->in[1] && len[0] % B == 0 && len[1] == j->n * 2 * B; break;
    case 2: j->n = len[0] / B; ok = j->in[0] && j->in[2] && j->in[3] && len[0] % B == 0 && len[2] == len[0] &&
                                     len[3] == j->n * 2 * B && (!j->in[1] || len[1] == j->n * 2 * B); break;
    case 3: ok = hl > 0 && j->in[0] && j->in[1] && j->in[2] && j->in[3] && len[0] % (size_t)hl == 0;
            j->n = ok ? len[0] / (size_t)hl : 0;
            ok = ok && len[1] == j->n * NB && len[2] == j->n * NB && len[3] == j->n * 2 * B; break;
    case 4: j->n = len[0] / 32; ok = j->in[0] && j->in[1] && len[0] % 32 == 0 && len[1] == len[0]; break;
    case 5: ok = hl > 0 && j->in[0] && j->in[1] && len[0] % (size_t)hl == 0;
            j->n = ok ? len[0] / (size_t)hl : 0;
            ok = ok && len[1] == j->n * NB; break;
    case 6: ok = hl > 0 && j->in[0] && j->in[1] && j->in[2] && j->in[3] && len[0] % (size_t)hl == 0;
            j->n = ok ? len[0] / (size_t)hl : 0;
            ok = ok && len[1] == j->n * NB && len[2] == j->n * NB && len[3] == j->n; break;
    case 7: ok = hl > 0 && i0 > 0 && i1 > 0 && j->in[0] && j->in[1] && j->in[2] && j->in[3] && len[0] % (size_t)hl == 0;
            j->n = ok ? len[0] / (size_t)hl : 0;
            ok = ok && len[1] == j->n * (size_t)i0 && len[2] == j->n * 4 && !((uintptr_t)j->in[2] & 3) &&
                 len[3] == j->n * (size_t)i1; break;
    default: ok = i0 > 0 && j->in[0] && len[0] % (size_t)i0 == 0;
             j->n = ok ? len[0] / (size_t)i0 : 0; break;
  }
  if (!ok) {
    for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
    free(j);
    THROW(env, "callAsync: buffer length mismatch");
  }
  j->out0_len = op == 3 || op == 7 ? j->n : op == 4 ? j->n * 32 : op == 5 ? j->n * NB : j->n * 2 * B;
  j->out1_len = op == 5 ? j->n * NB : j->n;
  j->out2_len = op == 5 ? j->n : 0;
  j->out3_len = op == 5 ? j->n : 0;
  j->out0 = (uint8_t*)malloc(j->out0_len ? j->out0_len : 1);
  j->out1 = (uint8_t*)malloc(j->out1_len ? j->out1_len : 1);
  j->out2 = (uint8_t*)malloc(j->out2_len ? j->out2_len : 1);
  j->out3 = (uint8_t*)malloc(j->out3_len ? j->out3_len : 1);
  if (!j->out0 || !j->out1 || !j->out2 || !j->out3) {
    for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
    free(j->out0); free(j->out1); free(j->out2); free(j->out3); free(j);
    THROW(env, "callAsync: out of memory");
  }
  napi_value promise, name;
  CHECK(env, napi_create_promise(env, &j->deferred, &promise));
  CHECK(env, napi_create_string_utf8(env, "ellgpu", NAPI_AUTO_LENGTH, &name));
  CHECK(env, napi_create_async_work(env, NULL, name, job_execute, job_complete, j, &j->work));
  CHECK(env, napi_queue_async_work(env, j->work));
  g_jobs_in_flight++;
  return promise;
}

static napi_value init(napi_env env, napi_value exports) {
  struct { const char* name; napi_callback fn; } fns[] = {
    {"open", fn_open}, {"createContext", fn_create}, {"destroyContext", fn_destroy},
    {"defer", fn_defer}, {"collect", fn_collect}, {"combBits", fn_comb_bits},
    {"curveId", fn_curve_id}, {"fieldBytes", fn_field_bytes}, {"orderBytes", fn_order_bytes},
    {"deviceCount", fn_device_count}, {"groupSize", fn_group_size}, {"defineShort", fn_define_short}, {"defineEdwards", fn_define_edwards}, {"mulFixed", fn_mul_fixed}, {"mulVar", fn_mul_var},
    {"mulAdd2", fn_mul_add2}, {"ecdsaVerify", fn_verify}, {"x25519", fn_x25519}, {"x25519Derive", fn_x25519_derive},
    {"callAsync", fn_call_async}, {"decompress", fn_decompress},
    {"eddsaVerify", fn_eddsa_verify}, {"eddsaSign", fn_eddsa_sign}, {"ecdsaSign", fn_sign}, {"ecdsaRecover", fn_recover}, {"ecdsaSignDet", fn_sign_det},
    {"decodePoints", fn_decode_points}, {"encodePoints", fn_encode_points}, {"validate", fn_validate},
    {"pointAdd", fn_point_add}, {"sigFromDer", fn_sig_from_der}, {"sigToDer", fn_sig_to_der}, {"ecdsaVerifyWire", fn_verify_wire},
  };
  napi_add_env_cleanup_hook(env, on_env_cleanup, NULL);
  for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
    if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
  }
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
