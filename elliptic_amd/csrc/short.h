// ellgpu -- short Weierstrass group law in Jacobian coordinates.
//
// Replaces the reference's JPoint (lib/elliptic/curve/short.js:482-938):
//   dbl        <- JPoint#dbl/_zeroDbl/_threeDbl (short.js:656-800)
//   add_mixed  <- JPoint#mixedAdd              (short.js:569-603)
//   add        <- JPoint#add                   (short.js:532-567)
// Only the AFFINE image of a result is canonical, so the
// formulas are the EFD ones best suited to a branch-free SIMT lane, not the
// reference's; the exceptional cases the reference branches on (operand at
// infinity, P == Q -> doubling, P == -Q -> infinity) are all honoured:
// infinity and P == -Q by selects, P == Q by a rarely-taken branch.
//
// A curve description CV supplies: F (field), A_KIND (0: a=0, 3: a=-3).
#pragma once

#include "fp.h"

namespace ell {

template <class F>
struct Jac {
  typename F::El X, Y, Z;      // Z == 0 <=> infinity
};

// 16-byte aligned: a table entry is fetched with dwordx4 loads (the 9-limb field's 72-byte
// entry is padded to 80 bytes: 5 loads instead of 18 single-dword ones)
template <class F>
struct alignas(16) Aff {
  typename F::El x, y;
};

template <class F>
ELL_HD typename F::El fe_select(bool c, const typename F::El& a, const typename F::El& b) {
  typename F::El r;
  bn_select<sizeof(typename F::El) / sizeof(u32)>(r.v, c, a.v, b.v);
  return r;
}

// F::LAZY (fpk256l.h): the field offers unnormalised add / sub and two-product multiplies; the
// group law below then takes its lazy formulas
template <class F, class = void>
struct is_lazy { static constexpr bool value = false; };
template <class F>
struct is_lazy<F, decltype((void)F::LAZY)> { static constexpr bool value = F::LAZY; };

// F::PAIR (fp.h FpSolinas): the field folds a difference of two wide products once
// (mul_sub_mul / mul_sub_sqr8); the doubling and the additions below form their Y3 that way
template <class F, class = void>
struct has_pair { static constexpr bool value = false; };
template <class F>
struct has_pair<F, decltype((void)F::PAIR)> { static constexpr bool value = F::PAIR; };

// F::QUAD (coop.h): a wave holds four field elements side by side and multiplies them in one
// instruction stream (F::mulq over F::pack* / F::unpack*); the lazy doubling and mixed addition
// below then take the independent products of a step together
template <class F, class = void>
struct has_quad { static constexpr bool value = false; };
template <class F>
struct has_quad<F, decltype((void)F::QUAD)> { static constexpr bool value = F::QUAD; };

template <class CV>
struct ShortOps {
  typedef typename CV::F F;
  typedef typename F::El El;
  typedef Jac<F> J;
  typedef Aff<F> A;

  ELL_HD static J infinity() {
    J r; r.X = F::one(); r.Y = F::one(); r.Z = F::zero(); return r;
  }
  ELL_HD static bool is_inf(const J& p) { return F::is_zero(p.Z); }
  ELL_HD static J from_affine(const A& q) {
    J r; r.X = q.x; r.Y = q.y; r.Z = F::one();
    // lazy fields: a table entry's y may carry an unnormalised conditional negation (Ladder::cneg_y)
    if constexpr (is_lazy<F>::value) r.Y = F::norm(q.y);
    return r;
  }
  ELL_HD static J select(bool c, const J& a, const J& b) {
    J r;
    r.X = fe_select<F>(c, a.X, b.X);
    r.Y = fe_select<F>(c, a.Y, b.Y);
    r.Z = fe_select<F>(c, a.Z, b.Z);
    return r;
  }
  ELL_HD static J cneg(const J& p, bool neg) {
    J r = p;
    r.Y = fe_select<F>(neg, F::neg(p.Y), p.Y);
    return r;
  }

  // 2P.  Z == 0 (infinity) and Y == 0 (order-2 point) both give Z3 == 0.
  // Lazy-field doubling, a = 0 (secp256k1 over FpK256L).  Formulas of libsecp256k1's gej_double
  // (the result is the usual one scaled by lambda = 1/2: same affine point), chosen because their
  // only constants are 3/2 and 2:
  //     L = 3/2 X^2,  S = Y^2,  T = X S,  X3 = L^2 - 2T,  Y3 = L (T - X3) - S^2,  Z3 = Y Z.
  // 3 S + 2 M + one two-product multiply; inputs and outputs in N form.
  template <class FF = F>
  ELL_HD static J dbl_lazy(const J& p) {
    J r;
    ELL_K256L_AT("dbl_lazy");
    if constexpr (has_quad<FF>::value) {
      // the same products in three steps: {Y^2, Y Z, X^2}, {X S, L^2}, the two-product Y3
      El s, a, t, l2;
      FF::unpack3(FF::mulq(FF::pack3(p.Y, p.Y, p.X), FF::pack3(p.Y, p.Z, p.X)), s, r.Z, a);
      El l = FF::norm(FF::add_l(a, FF::half_l(a)));
      FF::unpack2(FF::mulq(FF::pack2(p.X, l), FF::pack2(s, l)), t, l2);
      r.X = FF::norm(FF::template sub_l<4>(l2, FF::add_l(t, t)));
      El w = FF::template sub_l<4>(t, r.X);
      r.Y = FF::mul2(l, w, FF::template neg_l<2>(s), s);
      return r;
    }
    El s = FF::sqr(p.Y);                                             // N x N
    r.Z = FF::mul(p.Y, p.Z);
    El a = FF::sqr(p.X);
    // L = a + a / 2 (limbs below 2^30 + 2^28), normalised
    El l = FF::norm(FF::add_l(a, FF::half_l(a)));
    El t = FF::mul(p.X, s);
    // X3 = L^2 - 2T + 4p: limbs in (-2^30 - 2^19, 2^29 + 2^18), value > 0 (2T < 4p); an output -> N
    El l2 = FF::sqr(l);
    r.X = FF::norm(FF::template sub_l<4>(l2, FF::add_l(t, t)));
    // W = T - X3 + 4p (X3 is a norm output: < 2^257 + eps < 4p); Y3 = L W + (2p - S) S, one reduction
    El w = FF::template sub_l<4>(t, r.X);
    r.Y = FF::mul2(l, w, FF::template neg_l<2>(s), s);
    ELL_K256L_AT("after dbl_lazy");
    return r;
  }

  ELL_HD static J dbl(const J& p) {
    J r;
    if constexpr (is_lazy<F>::value && CV::A_KIND == 0) {
      return dbl_lazy<F>(p);
    } else if constexpr (CV::A_KIND == 1) {
      // dbl-2007-bl, arbitrary a (user-defined curves; the reference's JPoint#_dbl / dblp,
      // short.js:802-830, 605-654): 2M + 8S, a = F::curve_a() from the run-time parameter block
      El xx = F::sqr(p.X);
      El yy = F::sqr(p.Y);
      El yyyy = F::sqr(yy);
      El zz = F::sqr(p.Z);
      El s = F::template mul_pow2<1>(F::sub(F::sub(F::sqr(F::add(p.X, yy)), xx), yyyy));
      El m = F::add(F::add(F::template mul_pow2<1>(xx), xx), F::mul(F::curve_a(), F::sqr(zz)));
      r.X = F::sub(F::sqr(m), F::template mul_pow2<1>(s));
      r.Z = F::sub(F::sub(F::sqr(F::add(p.Y, p.Z)), yy), zz);
      r.Y = F::sub(F::mul(m, F::sub(s, r.X)), F::template mul_pow2<3>(yyyy));
      return r;
    } else if (CV::A_KIND == 0) {
      // dbl-2009-l, a = 0: 2M + 5S.  Statement order = shortest live ranges (Y and Z die
      // first, then X): at most five field elements are live next to a product's own words.
      El b = F::sqr(p.Y);
      r.Z = F::template mul_pow2<1>(F::mul(p.Y, p.Z));
      El a = F::sqr(p.X);
      El t = F::sub(F::sqr(F::add(p.X, b)), a);
      El c = F::sqr(b);
      El d = F::template mul_pow2<1>(F::sub(t, c));
      El e = F::add(F::template mul_pow2<1>(a), a);
      r.X = F::sub(F::sqr(e), F::template mul_pow2<1>(d));
      r.Y = F::sub(F::mul(e, F::sub(d, r.X)), F::template mul_pow2<3>(c));
    } else {
      // dbl-2001-b, a = -3: 3M + 5S
      if constexpr (has_quad<F>::value) {
        // the eight products in four steps: {Z^2, Y^2, (Y+Z)^2}, {X gamma, (X-delta)(X+delta), gamma^2}, alpha^2, Y3's
        El delta, gamma, yz, beta, t, g2;
        const El ypz = F::add(p.Y, p.Z);
        F::unpack3(F::mulq(F::pack3(p.Z, p.Y, ypz), F::pack3(p.Z, p.Y, ypz)), delta, gamma, yz);
        r.Z = F::sub(F::sub(yz, gamma), delta);
        F::unpack3(F::mulq(F::pack3(p.X, F::sub(p.X, delta), gamma), F::pack3(gamma, F::add(p.X, delta), gamma)), beta, t, g2);
        El alpha = F::add(F::template mul_pow2<1>(t), t);
        El beta4 = F::template mul_pow2<2>(beta);
        r.X = F::sub(F::sqr(alpha), F::template mul_pow2<1>(beta4));
        r.Y = F::sub(F::mul(alpha, F::sub(beta4, r.X)), F::template mul_pow2<3>(g2));
        return r;
      }
      El delta = F::sqr(p.Z);
      El gamma = F::sqr(p.Y);
      El beta = F::mul(p.X, gamma);
      El t = F::mul(F::sub(p.X, delta), F::add(p.X, delta));
      El alpha = F::add(F::template mul_pow2<1>(t), t);
      El beta4 = F::template mul_pow2<2>(beta);
      r.X = F::sub(F::sqr(alpha), F::template mul_pow2<1>(beta4));
      El yz = F::sqr(F::add(p.Y, p.Z));
      r.Z = F::sub(F::sub(yz, gamma), delta);
      if constexpr (has_pair<F>::value) {
        r.Y = F::mul_sub_sqr8(alpha, F::sub(beta4, r.X), gamma);
      } else {
        El g2 = F::sqr(gamma);
        El g8 = F::template mul_pow2<3>(g2);
        r.Y = F::sub(F::mul(alpha, F::sub(beta4, r.X)), g8);
      }
    }
    return r;
  }

  // P + Q, Q affine and finite; if !do_add returns P unchanged.  8M + 3S.
  ELL_HD static J add_mixed(const J& p, const A& q, bool do_add = true) {
    El z1z1 = F::sqr(p.Z);
    El u2 = F::mul(q.x, z1z1);
    El s2 = F::mul(q.y, F::mul(p.Z, z1z1));
    El h = F::sub(u2, p.X);
    El rr = F::sub(s2, p.Y);
    El hh = F::sqr(h);
    El hhh = F::mul(h, hh);
    El v = F::mul(p.X, hh);
    J r;
    r.X = F::sub(F::sub(F::sqr(rr), hhh), F::template mul_pow2<1>(v));
    r.Y = F::sub(F::mul(rr, F::sub(v, r.X)), F::mul(p.Y, hhh));
    r.Z = F::mul(p.Z, h);          // h == 0, rr != 0  ->  Z3 = 0: infinity
    // Every exceptional input (P = O, P = Q, P = -Q) gives Z3 = Z1 * h = 0: one zero test on
    // the common path, the case analysis behind a branch that is almost never taken.
    if (ELL_UNLIKELY(F::is_zero(r.Z))) {
      bool pinf = F::is_zero(p.Z);
      bool same = F::is_zero(h) && F::is_zero(rr) && !pinf;
      if (same) r = dbl(from_affine(q));                         // P == Q
      r = select(pinf, from_affine(q), r);                       // O + Q = Q  (P == -Q keeps Z3 = 0)
    }
    return select(do_add, r, p);
  }

  // P + Q for the ladders' inner loops, Q affine and finite.  Same formulas as add_mixed; what
  // differs is how the exceptional inputs are paid for.  add_mixed decides P = O / P = Q / P = -Q
  // from p, q and h at the end of the addition, which keeps all of them live through it (32 more
  // registers than the arithmetic needs).  Here the caller carries "P is O" as a flag (`pinf`,
  // updated on return -- an addition is the only step of a ladder that can produce or leave O),
  // Z3 = Z1 * h == 0 then means h == 0 exactly when the flag is clear, and all three exceptional
  // results depend on Q alone (Q, 2Q, O), which the rarely taken branch fetches again through
  // `reload` instead of holding it.  Statement order = shortest live ranges.
  // Lazy-field form of add_mixed_lean / add_mixed_zr (same formulas; p in N form, q = table entry
  // = direct product outputs): differences stay unnormalised with an offset K p that keeps their
  // VALUE positive, X3 is normalised once, Y3 = rr (v - X3) + (4p - Y1) hhh is one two-product
  // multiply.  3 S + 6 M + one two-product multiply.  `h_out` (optional) receives h.
  template <class FF = F>
  ELL_HD static J add_mixed_lazy(const J& p, const A& q, El* h_out, El* rr_out = nullptr) {
    ELL_K256L_AT("add_mixed_lazy");
    if constexpr (has_quad<FF>::value) {
      // five steps for the ten products: {Z^2, y2 Z}, {x2 Z^2, (y2 Z) Z^2}, {Z h, h^2, rr^2},
      // {h h^2, X h^2}, the two-product Y3  (s2 = (y2 Z) Z^2: the association differs from the
      // chain below, the value does not)
      El z1z1, yz, u2, s2, hh, r2, hhh, v;
      J r;
      FF::unpack2(FF::mulq(FF::pack2(p.Z, q.y), FF::pack2(p.Z, p.Z)), z1z1, yz);
      FF::unpack2(FF::mulq(FF::pack2(q.x, yz), FF::pack2(z1z1, z1z1)), u2, s2);
      El h = FF::template sub_l<4>(u2, p.X);
      El rr = FF::template sub_l<4>(s2, p.Y);
      FF::unpack3(FF::mulq(FF::pack3(p.Z, h, rr), FF::pack3(h, h, rr)), r.Z, hh, r2);
      FF::unpack2(FF::mulq(FF::pack2(h, p.X), FF::pack2(hh, hh)), hhh, v);
      r.X = FF::norm(FF::template sub_l<4>(r2, FF::add_l(hhh, FF::add_l(v, v))));
      El w = FF::template sub_l<4>(v, r.X);
      r.Y = FF::mul2(rr, w, FF::template neg_l<4>(p.Y), hhh);
      if (h_out) *h_out = FF::norm(h);
      if (rr_out) *rr_out = rr;
      return r;
    }
    El z1z1 = FF::sqr(p.Z);
    El u2 = FF::mul(q.x, z1z1);
    El s2 = FF::mul(q.y, FF::mul(p.Z, z1z1));
    El h = FF::template sub_l<4>(u2, p.X);                    // p.X, p.Y: N form (< 4p)
    J r;
    r.Z = FF::mul(p.Z, h);
    El rr = FF::template sub_l<4>(s2, p.Y);
    El hh = FF::sqr(h);
    El hhh = FF::mul(h, hh);
    El v = FF::mul(p.X, hh);
    // X3 = rr^2 - hhh - 2v + 4p  (three direct outputs, each < 2^256 + eps: their sum < 4p)
    El x3 = FF::template sub_l<4>(FF::sqr(rr), FF::add_l(hhh, FF::add_l(v, v)));
    r.X = FF::norm(x3);
    El w = FF::template sub_l<4>(v, r.X);                     // v - X3 + 4p
    r.Y = FF::mul2(rr, w, FF::template neg_l<4>(p.Y), hhh);
    if (h_out) *h_out = FF::norm(h);
    if (rr_out) *rr_out = rr;
    ELL_K256L_AT("after add_mixed_lazy");
    return r;
  }

  // the eleven products of the mixed addition in five steps (fields with F::QUAD, normalising
  // interface): {Z^2, y2 Z}, {x2 Z^2, (y2 Z) Z^2}, {Z h, h^2, rr^2}, {h h^2, X h^2}, {rr (v - X3), Y h^3}
  ELL_HD static J add_mixed_quad(const J& p, const A& q, El& h, El& rr) {
    El z1z1, yz, u2, s2, hh, r2, hhh, v, ya, yb;
    J r;
    F::unpack2(F::mulq(F::pack2(p.Z, q.y), F::pack2(p.Z, p.Z)), z1z1, yz);
    F::unpack2(F::mulq(F::pack2(q.x, yz), F::pack2(z1z1, z1z1)), u2, s2);
    h = F::sub(u2, p.X);
    rr = F::sub(s2, p.Y);
    F::unpack3(F::mulq(F::pack3(p.Z, h, rr), F::pack3(h, h, rr)), r.Z, hh, r2);
    F::unpack2(F::mulq(F::pack2(h, p.X), F::pack2(hh, hh)), hhh, v);
    r.X = F::sub(F::sub(r2, hhh), F::template mul_pow2<1>(v));
    F::unpack2(F::mulq(F::pack2(rr, p.Y), F::pack2(F::sub(v, r.X), hhh)), ya, yb);
    r.Y = F::sub(ya, yb);
    return r;
  }

  template <class Reload>
  ELL_HD static J add_mixed_lean(const J& p, const A& q, bool& pinf, const Reload& reload) {
    if constexpr (is_lazy<F>::value) {
      El rr;
      J r = add_mixed_lazy<F>(p, q, nullptr, &rr);
      bool z = F::is_zero_w(r.Z);
      if (ELL_UNLIKELY(z)) {
        const A qq = reload();
        // Z3 = Z1 h = 0 with P finite means h = 0; then rr = S2 - Y1 (still at hand, lazy: the
        // canonical zero test takes any non-negative lazy value) decides P == Q / P == -Q
        const bool same = !pinf && F::is_zero(rr);
        if (pinf) r = from_affine(qq);
        else if (same) r = dbl(from_affine(qq));
        z = F::is_zero(r.Z);
      }
      pinf = z;
      return r;
    }
    J r;
    El rr;
    if constexpr (has_quad<F>::value) {
      El h;
      r = add_mixed_quad(p, q, h, rr);
    } else {
      El z1z1 = F::sqr(p.Z);
      El u2 = F::mul(q.x, z1z1);
      El s2 = F::mul(q.y, F::mul(p.Z, z1z1));
      El h = F::sub(u2, p.X);
      r.Z = F::mul(p.Z, h);
      rr = F::sub(s2, p.Y);
      El hh = F::sqr(h);
      El hhh = F::mul(h, hh);
      El v = F::mul(p.X, hh);
      r.X = F::sub(F::sub(F::sqr(rr), hhh), F::template mul_pow2<1>(v));
      if constexpr (has_pair<F>::value) {
        r.Y = F::mul_sub_mul(rr, F::sub(v, r.X), p.Y, hhh);
      } else {
        El yh = F::mul(p.Y, hhh);
        r.Y = F::sub(F::mul(rr, F::sub(v, r.X)), yh);
      }
    }
    bool z = F::is_zero(r.Z);
    if (ELL_UNLIKELY(z)) {
      const A qq = reload();
      const bool same = !pinf && F::is_zero(rr);                 // h == 0 and rr == 0: P == Q
      if (pinf) r = from_affine(qq);                              // O + Q = Q
      else if (same) r = dbl(from_affine(qq));                    // (P == -Q keeps Z3 = 0)
      z = F::is_zero(r.Z);
    }
    pinf = z;
    return r;
  }

  // P + Q for table building: no exceptional cases are possible (P = j*Q0, Q = 2*Q0 on a
  // prime-order curve), returns the ratio h with Z3 = Z1 * h.  8M + 3S.
  ELL_HD static J add_mixed_zr(const J& p, const A& q, El& h) {
    if constexpr (is_lazy<F>::value) return add_mixed_lazy<F>(p, q, &h);
    if constexpr (has_quad<F>::value) {
      El rr;
      return add_mixed_quad(p, q, h, rr);
    }
    El z1z1 = F::sqr(p.Z);
    El u2 = F::mul(q.x, z1z1);
    El s2 = F::mul(q.y, F::mul(p.Z, z1z1));
    h = F::sub(u2, p.X);
    El rr = F::sub(s2, p.Y);
    El hh = F::sqr(h);
    El hhh = F::mul(h, hh);
    El v = F::mul(p.X, hh);
    J r;
    r.X = F::sub(F::sub(F::sqr(rr), hhh), F::template mul_pow2<1>(v));
    r.Y = F::sub(F::mul(rr, F::sub(v, r.X)), F::mul(p.Y, hhh));
    r.Z = F::mul(p.Z, h);
    return r;
  }

  // P + Q, both Jacobian; if !do_add returns P unchanged.  12M + 4S.
  ELL_HD static J add(const J& p, const J& q, bool do_add = true) {
    El z1z1 = F::sqr(p.Z);
    El z2z2 = F::sqr(q.Z);
    El u1 = F::mul(p.X, z2z2);
    El u2 = F::mul(q.X, z1z1);
    El s1 = F::mul(p.Y, F::mul(q.Z, z2z2));
    El s2 = F::mul(q.Y, F::mul(p.Z, z1z1));
    El h = F::sub(u2, u1);
    El rr = F::sub(s2, s1);
    El hh = F::sqr(h);
    El hhh = F::mul(h, hh);
    El v = F::mul(u1, hh);
    J r;
    r.X = F::sub(F::sub(F::sqr(rr), hhh), F::template mul_pow2<1>(v));
    r.Y = F::sub(F::mul(rr, F::sub(v, r.X)), F::mul(s1, hhh));
    r.Z = F::mul(F::mul(p.Z, q.Z), h);
    // P = O, Q = O, P = Q and P = -Q all give Z3 = Z1 * Z2 * h = 0
    if (ELL_UNLIKELY(F::is_zero(r.Z))) {
      bool pinf = F::is_zero(p.Z);
      bool qinf = F::is_zero(q.Z);
      bool same = F::is_zero(h) && F::is_zero(rr) && !pinf && !qinf;
      if (same) r = dbl(p);                                      // P == Q
      r = select(pinf, q, r);                                    // O + Q = Q
      r = select(qinf, p, r);                                    // P + O = P
    }
    return select(do_add, r, p);
  }
};

}  // namespace ell
