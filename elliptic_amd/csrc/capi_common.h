// ellgpu -- the extern "C" surface of include/ellgpu.h, written once over
// Engine<ELL_BACKEND>.  Included by capi.hip (ELL_BACKEND = HipBackend, the
// product) and by tests/hostsim/hostsim.cpp (ELL_BACKEND = LoopBackend, CPU
// unit-test build of the same code).  The including file must define
//   ELL_BACKEND            backend type
//   ell_backend_create(int device, ELL_BACKEND* out, std::string* err) -> int
//   ell_backend_destroy(ELL_BACKEND*)
//   ell_device_count() -> int
// before including this header.
#pragma once

#include "../../include/ellgpu.h"
#include "engine.h"

#include <mutex>
#include <thread>
#include <vector>

// A context owns one engine on one device.  A GROUP (ellgpu_group_create) is a context whose
// `members` are ordinary contexts, one per listed device: the host-buffer scalar-multiplication
// and verify entry points cut a batch into contiguous shards, one host thread per member drives
// its device (its own streams, scratch arena and replicated tables), and every member copies its
// results straight into the caller's buffers -- no device-to-device traffic at all (SURVEY.md 8e:
// "a direct per-device D2H is the zero-collective alternative").  Any other entry point called on
// a group runs on member 0.
// `mu` serialises the entry points of ONE context: the staging buffers, the scratch arenas, the
// stream bookkeeping and `err` belong to the context, so two host threads inside it at once -- the
// N-API addon's libuv worker running a Promise-form batch while the JS thread makes a synchronous
// call -- would overwrite each other's staged inputs.  The second caller waits (host-buffer calls
// hold the lock until their results are back; *_dev calls only while they enqueue).  Recursive:
// an entry point may call another on the same context.  A group call locks the group, then every
// member from its worker thread.
struct ellgpu_ctx {
  ell::Engine<ELL_BACKEND>* eng;
  std::vector<ellgpu_ctx*> members;
  std::recursive_mutex mu;
  int deferred_rc = 0;             // status of a deferred call that something else completed
  std::string deferred_msg;
};

// (entering a context also completes a call that ellgpu_ctx_defer left in flight: its results sit
// in the context's one pinned buffer, which the next small call would overwrite)
static void ell_complete_deferred(ellgpu_ctx* ctx) {
  if (!ctx->members.empty() || !ctx->eng || !ctx->eng->defer_pending()) return;
  const int rc = ctx->eng->defer_collect();
  if (rc && !ctx->deferred_rc) { ctx->deferred_rc = rc; ctx->deferred_msg = ctx->eng->err; }
  ctx->eng->bk.end_call(false);
}
#define ELL_LOCK(ctx) std::lock_guard<std::recursive_mutex> ell_ctx_lock_((ctx)->mu); ell_complete_deferred(ctx)

static thread_local std::string g_last_error;

static int set_err(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
// end of every entry point.  `stream_call` = a *_dev entry point: it returns without
// synchronising, so the backend records where its work ends (see HipBackend::use_stream).
static int finish(ellgpu_ctx* ctx, int rc, bool stream_call = false) {
  ctx->eng->bk.end_call(stream_call);
  if (rc) g_last_error = ctx->eng->err.empty() ? "ellgpu error" : ctx->eng->err;
  return rc;
}

extern "C" {

int ellgpu_version(void) { return ELLGPU_VERSION; }
#ifndef ELLGPU_SOURCE_DIGEST
#define ELLGPU_SOURCE_DIGEST "unstamped"
#endif
const char* ellgpu_source_digest(void) { return ELLGPU_SOURCE_DIGEST; }
const char* ellgpu_last_error(void) { return g_last_error.c_str(); }

int ellgpu_curve_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < ell::CURVE_COUNT; i++)
    if (strcmp(name, ell::curve_info(i)->name) == 0) return i;
  return -1;
}
int ellgpu_curve_field_bytes(int curve) {
  const ell::CurveInfo* ci = ell::curve_info(curve);
  return ci ? ci->field_bytes : ELLGPU_E_ARG;
}
int ellgpu_curve_order_bytes(int curve) {
  const ell::CurveInfo* ci = ell::curve_info(curve);
  return ci ? ci->order_bytes : ELLGPU_E_ARG;
}
int ellgpu_device_count(void) { return ell_device_count(); }

int ellgpu_ctx_create(int device, ellgpu_ctx** out) {
  if (!out) return set_err(ELLGPU_E_ARG, "null out pointer");
  *out = nullptr;
  ELL_BACKEND bk;
  std::string err;
  int rc = ell_backend_create(device, &bk, &err);
  if (rc) return set_err(rc, err);
  ellgpu_ctx* c = new ellgpu_ctx;
  c->eng = new ell::Engine<ELL_BACKEND>(bk);
  *out = c;
  return ELLGPU_OK;
}
int ellgpu_group_create(const int* devices, int ndev, ellgpu_ctx** out) {
  if (!out) return set_err(ELLGPU_E_ARG, "null out pointer");
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > 64) return set_err(ELLGPU_E_ARG, "ellgpu_group_create: 1..64 devices");
  ellgpu_ctx* g = new ellgpu_ctx;
  g->eng = nullptr;
  for (int i = 0; i < ndev; i++) {
    ellgpu_ctx* m = nullptr;
    int rc = ellgpu_ctx_create(devices[i], &m);
    if (rc) {
      for (ellgpu_ctx* x : g->members) ellgpu_ctx_destroy(x);
      delete g;
      return rc;                                    // message already set by ellgpu_ctx_create
    }
    g->members.push_back(m);
  }
  g->eng = g->members[0]->eng;                      // borrowed: entry points without a sharded form
  *out = g;
  return ELLGPU_OK;
}
int ellgpu_group_size(const ellgpu_ctx* ctx) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  return ctx->members.empty() ? 1 : (int)ctx->members.size();
}
}  // extern "C"
// run f(member, lo, hi) for the contiguous shard of every member on its own host thread
template <class Fn>
static int group_shard(ellgpu_ctx* g, size_t n, Fn f) {
  const size_t m = g->members.size();
  std::vector<int> rc(m, 0);
  std::vector<std::string> msg(m);
  std::vector<std::thread> th;
  for (size_t i = 0; i < m; i++) {
    const size_t lo = n * i / m, hi = n * (i + 1) / m;
    th.emplace_back([&, i, lo, hi]() {
      rc[i] = lo < hi ? f(g->members[i], lo, hi) : 0;
      if (rc[i]) msg[i] = g_last_error;             // the worker's thread-local message
    });
  }
  for (auto& t : th) t.join();
  for (size_t i = 0; i < m; i++)
    if (rc[i]) return set_err(rc[i], "device shard " + std::to_string(i) + ": " + msg[i]);
  return ELLGPU_OK;
}
extern "C" {

void ellgpu_ctx_destroy(ellgpu_ctx* ctx) {
  if (!ctx) return;
  if (!ctx->members.empty()) {
    for (ellgpu_ctx* m : ctx->members) ellgpu_ctx_destroy(m);
    delete ctx;
    return;
  }
  ctx->eng->bk.sync();
  ELL_BACKEND bk = ctx->eng->bk;
  delete ctx->eng;
  ell_backend_destroy(&bk);
  delete ctx;
}
int ellgpu_ctx_synchronize(ellgpu_ctx* ctx) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  ELL_LOCK(ctx);
  if (!ctx->members.empty()) {                       // a group: every member's device
    for (ellgpu_ctx* m : ctx->members) {
      int rc = ellgpu_ctx_synchronize(m);
      if (rc) return rc;
    }
    return ELLGPU_OK;
  }
  return finish(ctx, ctx->eng->bk.sync());
}
void* ellgpu_ctx_stream(ellgpu_ctx* ctx) {
  if (!ctx) return nullptr;
  ELL_LOCK(ctx);
  return ctx->eng->bk.own_stream();
}
int ellgpu_ctx_defer(ellgpu_ctx* ctx) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  ELL_LOCK(ctx);
  if (ctx->members.empty()) ctx->eng->defer_arm();          // (a group's calls are never deferred)
  return ELLGPU_OK;
}
int ellgpu_ctx_collect(ellgpu_ctx* ctx) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  ELL_LOCK(ctx);                                             // completes the pending call, if any
  if (ctx->members.empty()) ctx->eng->defer_collect();       // disarms
  const int rc = ctx->deferred_rc;
  ctx->deferred_rc = 0;
  if (rc) return set_err(rc, ctx->deferred_msg.empty() ? "device error in a deferred call" : ctx->deferred_msg);
  return ELLGPU_OK;
}
int ellgpu_ctx_reserve(ellgpu_ctx* ctx, int curve, size_t n) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  ELL_LOCK(ctx);
  if (!ctx->members.empty()) {
    // a group: every member gets its shard's worth (tables and scratch are per device), so that
    // the first sharded call does not allocate and build tables inside its worker threads
    const size_t m = ctx->members.size();
    return group_shard(ctx, m, [=](ellgpu_ctx* mem, size_t lo, size_t) {
      return ellgpu_ctx_reserve(mem, curve, (n * (lo + 1) / m) - (n * lo / m));
    });
  }
  ctx->eng->bk.use_stream(nullptr);
  return finish(ctx, ctx->eng->reserve(curve, n));
}

int ellgpu_ctx_comb_bits(ellgpu_ctx* ctx, int curve) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  if (curve < 0 || curve >= ell::CURVE_COUNT) return set_err(ELLGPU_E_ARG, "unknown curve id");
  ELL_LOCK(ctx);
  return ctx->eng->comb_bits(curve);
}

static int define_custom(ellgpu_ctx* ctx, int edwards, const uint8_t* p, const uint8_t* a, const uint8_t* b,
                         int* out_curve) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  ELL_LOCK(ctx);
  if (!ctx->members.empty()) {
    // all or nothing: a member that cannot take the curve (table full), or would give it another
    // id than member 0 (curves were defined on a member directly), is found BEFORE anything is
    // registered -- the engine dedups identical parameters, so the probe is the definition's own
    // first half (Engine::custom_slot_for)
    int id = -1;
    for (size_t i = 0; i < ctx->members.size(); i++) {
      int mid = ctx->members[i]->eng->custom_slot_for(edwards, p, a, b);
      if (mid < 0) return set_err(ELLGPU_E_UNSUPPORTED, "user-defined curve table of a group member is full");
      if (i && mid != id) return set_err(ELLGPU_E_ARG, "group members disagree on the curve id (curves were defined on a member directly)");
      id = mid;
    }
    for (size_t i = 0; i < ctx->members.size(); i++) {
      int mid = -1;
      int rc = define_custom(ctx->members[i], edwards, p, a, b, &mid);
      if (rc) return rc;                             // (parameter errors are the same on every member: member 0 fails first)
    }
    if (out_curve) *out_curve = id;
    return ELLGPU_OK;
  }
  ctx->eng->err.clear();
  int rc = edwards ? ctx->eng->define_edwards(p, a, b, out_curve) : ctx->eng->define_short(p, a, b, out_curve);
  if (rc) g_last_error = ctx->eng->err.empty() ? "ellgpu error" : ctx->eng->err;
  return rc;
}
int ellgpu_curve_define_short(ellgpu_ctx* ctx, const uint8_t* p, const uint8_t* a, const uint8_t* b,
                              int* out_curve) {
  return define_custom(ctx, 0, p, a, b, out_curve);
}
int ellgpu_curve_define_edwards(ellgpu_ctx* ctx, const uint8_t* p, const uint8_t* a, const uint8_t* d,
                                int* out_curve) {
  return define_custom(ctx, 1, p, a, d, out_curve);
}

#define ELL_ENTER(ctx, stream)                                      \
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");           \
  ELL_LOCK(ctx);                                                    \
  ctx->eng->err.clear();                                            \
  ctx->eng->bk.use_stream(stream);                                  \
  ctx->eng->set_lane(0);
// *_dev entry points: the call works in the scratch arena of its stream (HipBackend::use_stream_dev)
#define ELL_ENTER_DEV(ctx, stream)                                  \
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");           \
  ELL_LOCK(ctx);                                                    \
  ctx->eng->err.clear();                                            \
  ctx->eng->set_lane(ctx->eng->bk.use_stream_dev(stream));

// byte widths of a curve's field elements and scalars (for slicing the flat buffers of a group call)
static int curve_widths(int curve, size_t& B, size_t& NB) {
  const ell::CurveInfo* ci = ell::curve_info(curve);
  if (!ci) return set_err(ELLGPU_E_ARG, "unknown curve id");
  B = (size_t)ci->field_bytes;
  NB = (size_t)ci->order_bytes;
  return 0;
}

int ellgpu_mul_fixed(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k, uint8_t* out_xy,
                     uint8_t* out_inf) {
  if (ctx && !ctx->members.empty()) {
    ELL_LOCK(ctx);
    size_t B, NB;
    if (curve_widths(curve, B, NB)) return ELLGPU_E_ARG;
    if (n && (!k || !out_xy)) return set_err(ELLGPU_E_ARG, "null buffer");
    return group_shard(ctx, n, [=](ellgpu_ctx* m, size_t lo, size_t hi) {
      return ellgpu_mul_fixed(m, curve, hi - lo, k + lo * B, out_xy + lo * 2 * B, out_inf ? out_inf + lo : nullptr);
    });
  }
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->mul_fixed_host(curve, n, k, out_xy, out_inf));
}
int ellgpu_mul_var(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k, const uint8_t* in_xy,
                   uint8_t* out_xy, uint8_t* out_inf) {
  if (ctx && !ctx->members.empty()) {
    ELL_LOCK(ctx);
    size_t B, NB;
    if (curve_widths(curve, B, NB)) return ELLGPU_E_ARG;
    if (n && (!k || !in_xy || !out_xy)) return set_err(ELLGPU_E_ARG, "null buffer");
    return group_shard(ctx, n, [=](ellgpu_ctx* m, size_t lo, size_t hi) {
      return ellgpu_mul_var(m, curve, hi - lo, k + lo * B, in_xy + lo * 2 * B, out_xy + lo * 2 * B,
                            out_inf ? out_inf + lo : nullptr);
    });
  }
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->mul_var_host(curve, n, k, in_xy, out_xy, out_inf));
}
int ellgpu_mul_add2(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k1, const uint8_t* p1_xy,
                    const uint8_t* k2, const uint8_t* p2_xy, uint8_t* out_xy, uint8_t* out_inf) {
  if (ctx && !ctx->members.empty()) {
    ELL_LOCK(ctx);
    size_t B, NB;
    if (curve_widths(curve, B, NB)) return ELLGPU_E_ARG;
    if (n && (!k1 || !k2 || !p2_xy || !out_xy)) return set_err(ELLGPU_E_ARG, "null buffer");
    return group_shard(ctx, n, [=](ellgpu_ctx* m, size_t lo, size_t hi) {
      return ellgpu_mul_add2(m, curve, hi - lo, k1 + lo * B, p1_xy ? p1_xy + lo * 2 * B : nullptr, k2 + lo * B,
                             p2_xy + lo * 2 * B, out_xy + lo * 2 * B, out_inf ? out_inf + lo : nullptr);
    });
  }
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->mul_add2_host(curve, n, k1, p1_xy, k2, p2_xy, out_xy, out_inf));
}
int ellgpu_ecdsa_verify(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                        int msg_bits, const uint8_t* r, const uint8_t* s, const uint8_t* pub_xy,
                        uint8_t* out_ok, uint8_t* out_status) {
  if (ctx && !ctx->members.empty()) {
    ELL_LOCK(ctx);
    size_t B, NB;
    if (curve_widths(curve, B, NB)) return ELLGPU_E_ARG;
    if (n && (!hash || !r || !s || !pub_xy || !out_ok)) return set_err(ELLGPU_E_ARG, "null buffer");
    return group_shard(ctx, n, [=](ellgpu_ctx* m, size_t lo, size_t hi) {
      return ellgpu_ecdsa_verify(m, curve, hi - lo, hash + lo * (size_t)hash_len, hash_len, msg_bits, r + lo * NB,
                                 s + lo * NB, pub_xy + lo * 2 * B, out_ok + lo,
                                 out_status ? out_status + lo : nullptr);
    });
  }
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->ecdsa_verify_host(curve, n, hash, hash_len, msg_bits, r, s, pub_xy,
                                                 out_ok, out_status));
}
int ellgpu_x25519_ladder(ellgpu_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* in_x,
                         uint8_t* out_x, uint8_t* out_inf) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->x25519_host(n, k, in_x, out_x, out_inf));
}

int ellgpu_x25519_derive(ellgpu_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* in_x,
                         uint8_t* out_x, uint8_t* out_status) {
  ELL_ENTER(ctx, nullptr);
  if (n && !out_status) return set_err(ELLGPU_E_ARG, "null buffer");
  // (never deferred: the status bytes are put together on the host)
  ctx->eng->defer_collect();
  std::vector<uint8_t> inf(n ? n : 1);
  const int rc = ctx->eng->x25519_host(n, k, in_x, out_x, inf.data(), out_status);
  if (!rc)
    for (size_t i = 0; i < n; i++) out_status[i] = out_status[i] ? 1 : (inf[i] ? 2 : 0);
  return finish(ctx, rc);
}

int ellgpu_decompress(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* v, const uint8_t* odd,
                      uint8_t* out_xy, uint8_t* out_ok) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->decompress_host(curve, n, v, odd, out_xy, out_ok));
}
int ellgpu_decompress_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* v, const uint8_t* odd,
                          uint8_t* out_xy, uint8_t* out_ok, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->decompress_dev(curve, n, v, odd, out_xy, out_ok), true);
}

// point codecs and key validation (decodePoint / encode / KeyPair#validate)
int ellgpu_decode_points(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* enc, size_t enc_len,
                         uint8_t* out_xy, uint8_t* out_status) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->decode_points_host(curve, n, enc, enc_len, out_xy, out_status));
}
int ellgpu_decode_points_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* enc, size_t enc_len,
                             uint8_t* out_xy, uint8_t* out_status, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->decode_points_dev(curve, n, enc, enc_len, out_xy, out_status), true);
}
int ellgpu_encode_points(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, int compact,
                         uint8_t* out_enc) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->encode_points_host(curve, n, xy, compact, out_enc));
}
int ellgpu_encode_points_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, int compact,
                             uint8_t* out_enc, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->encode_points_dev(curve, n, xy, compact, out_enc), true);
}
int ellgpu_validate(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, const uint8_t* inf,
                    int check_order, uint8_t* out_status) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->validate_host(curve, n, xy, inf, check_order, out_status));
}
int ellgpu_validate_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, const uint8_t* inf,
                        int check_order, uint8_t* out_status, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->validate_dev(curve, n, xy, inf, check_order, out_status), true);
}

// Point#add on affine points
int ellgpu_point_add(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy1, const uint8_t* inf1,
                     const uint8_t* xy2, const uint8_t* inf2, uint8_t* out_xy, uint8_t* out_inf) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->point_add_host(curve, n, xy1, inf1, xy2, inf2, out_xy, out_inf));
}
int ellgpu_point_add_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy1, const uint8_t* inf1,
                         const uint8_t* xy2, const uint8_t* inf2, uint8_t* out_xy, uint8_t* out_inf,
                         void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->point_add_dev(curve, n, xy1, inf1, xy2, inf2, out_xy, out_inf), true);
}

// signature DER codec and EC#verify on wire formats
int ellgpu_sig_from_der(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* der, size_t stride,
                        const uint32_t* der_len, uint8_t* out_r, uint8_t* out_s, uint8_t* out_status) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->sig_from_der_host(curve, n, der, stride, der_len, out_r, out_s, out_status));
}
int ellgpu_sig_from_der_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* der, size_t stride,
                            const uint32_t* der_len, uint8_t* out_r, uint8_t* out_s, uint8_t* out_status,
                            void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->sig_from_der_dev(curve, n, der, stride, der_len, out_r, out_s, out_status), true);
}
int ellgpu_sig_to_der(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* r, const uint8_t* s,
                      uint8_t* out_der, size_t stride, uint32_t* out_len) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->sig_to_der_host(curve, n, r, s, out_der, stride, out_len));
}
int ellgpu_sig_to_der_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* r, const uint8_t* s,
                          uint8_t* out_der, size_t stride, uint32_t* out_len, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->sig_to_der_dev(curve, n, r, s, out_der, stride, out_len), true);
}
int ellgpu_ecdsa_verify_wire(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                             int msg_bits, const uint8_t* der, size_t der_stride, const uint32_t* der_len,
                             const uint8_t* pub_enc, size_t pub_len, uint8_t* out_ok, uint8_t* out_err) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->ecdsa_verify_wire_host(curve, n, hash, hash_len, msg_bits, der, der_stride,
                                                      der_len, pub_enc, pub_len, out_ok, out_err));
}
int ellgpu_ecdsa_verify_wire_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                                 int msg_bits, const uint8_t* der, size_t der_stride,
                                 const uint32_t* der_len, const uint8_t* pub_enc, size_t pub_len,
                                 uint8_t* out_ok, uint8_t* out_err, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->ecdsa_verify_wire_dev(curve, n, hash, hash_len, msg_bits, der, der_stride,
                                                     der_len, pub_enc, pub_len, out_ok, out_err), true);
}

int ellgpu_ecdsa_sign(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len, int msg_bits,
                      const uint8_t* priv, const uint8_t* nonces, int canonical, uint8_t* out_r,
                      uint8_t* out_s, uint8_t* out_recid, uint8_t* out_ok) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->ecdsa_sign_host(curve, n, hash, hash_len, msg_bits, priv, nonces, canonical,
                                               out_r, out_s, out_recid, out_ok));
}
int ellgpu_ecdsa_sign_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                          int msg_bits, const uint8_t* priv, const uint8_t* nonces, int canonical,
                          uint8_t* out_r, uint8_t* out_s, uint8_t* out_recid, uint8_t* out_ok,
                          void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->ecdsa_sign_dev(curve, n, hash, hash_len, msg_bits, priv, nonces, canonical,
                                              out_r, out_s, out_recid, out_ok), true);
}

int ellgpu_eddsa_verify(ellgpu_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off,
                        size_t msg_len, const uint8_t* sigs, const uint8_t* pubs, uint8_t* out_ok,
                        uint8_t* out_err) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->eddsa_verify_host(n, msgs, (const ell::u64*)msg_off, msg_len, sigs, pubs,
                                                 out_ok, out_err));
}
int ellgpu_eddsa_verify_dev(ellgpu_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off,
                            size_t msg_len, const uint8_t* sigs, const uint8_t* pubs,
                            uint8_t* out_ok, uint8_t* out_err, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->eddsa_verify_dev(n, msgs, (const ell::u64*)msg_off, msg_len, sigs, pubs,
                                                out_ok, out_err), true);
}

int ellgpu_ecdsa_sign_det(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                          int msg_bits, const uint8_t* priv, int canonical, uint8_t* out_r, uint8_t* out_s,
                          uint8_t* out_recid, uint8_t* out_ok) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->ecdsa_sign_det_host(curve, n, hash, hash_len, msg_bits, priv, canonical, out_r,
                                                   out_s, out_recid, out_ok));
}
int ellgpu_ecdsa_sign_det_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                              int msg_bits, const uint8_t* priv, int canonical, uint8_t* out_r,
                              uint8_t* out_s, uint8_t* out_recid, uint8_t* out_ok, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->ecdsa_sign_det_dev(curve, n, hash, hash_len, msg_bits, priv, canonical, out_r,
                                                  out_s, out_recid, out_ok), true);
}

int ellgpu_ecdsa_recover(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                         const uint8_t* r, const uint8_t* s, const uint8_t* recid, uint8_t* out_xy,
                         uint8_t* out_status) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->ecdsa_recover_host(curve, n, hash, hash_len, r, s, recid, out_xy, out_status));
}
int ellgpu_ecdsa_recover_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                             const uint8_t* r, const uint8_t* s, const uint8_t* recid, uint8_t* out_xy,
                             uint8_t* out_status, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->ecdsa_recover_dev(curve, n, hash, hash_len, r, s, recid, out_xy, out_status), true);
}

int ellgpu_eddsa_sign(ellgpu_ctx* ctx, size_t n, const uint8_t* secrets, const uint8_t* msgs,
                      const uint64_t* msg_off, size_t msg_len, uint8_t* out_sig, uint8_t* out_pub) {
  ELL_ENTER(ctx, nullptr);
  return finish(ctx, ctx->eng->eddsa_sign_host(n, secrets, msgs, (const ell::u64*)msg_off, msg_len, out_sig,
                                               out_pub));
}
int ellgpu_eddsa_sign_dev(ellgpu_ctx* ctx, size_t n, const uint8_t* secrets, const uint8_t* msgs,
                          const uint64_t* msg_off, size_t msg_len, uint8_t* out_sig, uint8_t* out_pub,
                          void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->eddsa_sign_dev(n, secrets, msgs, (const ell::u64*)msg_off, msg_len, out_sig,
                                              out_pub), true);
}

int ellgpu_mul_fixed_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k, uint8_t* out_xy,
                         uint8_t* out_inf, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->mul_fixed_dev(curve, n, k, out_xy, out_inf), true);
}
int ellgpu_mul_var_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k,
                       const uint8_t* in_xy, uint8_t* out_xy, uint8_t* out_inf, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->mul_var_dev(curve, n, k, in_xy, out_xy, out_inf), true);
}
int ellgpu_mul_add2_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k1,
                        const uint8_t* p1_xy, const uint8_t* k2, const uint8_t* p2_xy,
                        uint8_t* out_xy, uint8_t* out_inf, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->mul_add2_dev(curve, n, k1, p1_xy, k2, p2_xy, out_xy, out_inf), true);
}
int ellgpu_ecdsa_verify_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash,
                            int hash_len, int msg_bits, const uint8_t* r, const uint8_t* s,
                            const uint8_t* pub_xy, uint8_t* out_ok, uint8_t* out_status,
                            void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->ecdsa_verify_dev(curve, n, hash, hash_len, msg_bits, r, s, pub_xy,
                                                out_ok, out_status), true);
}
int ellgpu_x25519_ladder_dev(ellgpu_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* in_x,
                             uint8_t* out_x, uint8_t* out_inf, void* stream) {
  ELL_ENTER_DEV(ctx, stream);
  return finish(ctx, ctx->eng->x25519_dev(n, k, in_x, out_x, out_inf), true);
}

}  // extern "C"
