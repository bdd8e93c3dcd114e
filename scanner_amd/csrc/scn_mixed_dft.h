// scn_mixed_dft.h -- in-register forward DFTs of any length R = 2^a 3^b 5^c (R <= 32 in use) for the mixed-radix fused kernels
// (scn_mixed.hip): fft.cpp:4-11 plans whatever --count it is given, and FFTW runs the 5-smooth sizes (1000, 6000, 12000 ...) with
// the same kind of codelets.  Plain arithmetic on a complex type with members x, y -- no device builtins -- so that the same
// header compiles for the host: tests/cpp/test_mixed_dft.cpp holds every length against the DFT sum in double on the CPU.
//
// The complex type decides the precision: members of type float (cf) or double (cd: the last pass of the sizes beyond 10000, where a
// strong tone's float partial sums put the parity metric's tail on the bar -- scn_mixed.hip); the constants follow it.
//
// scn_dft<R>(v): in place, natural order in and out.  R in {2, 3, 4, 5} are straight-line butterflies; any other length is one
// Cooley-Tukey step R = A B (A = 4 where that divides, else the smallest prime factor), n = B a + b, k = p + A q:
//   y[p][b] = W_R^(b p) * sum_a v[B a + b] W_A^(a p);   X[p + A q] = sum_b y[p][b] W_B^(b q)
// fully unrolled: every index and every twiddle is a compile-time constant, the arrays live in registers.
#pragma once

#if defined(__HIPCC__)
#define SCN_DFT_FN __device__ __forceinline__
#else
#define SCN_DFT_FN inline
#endif

template <class C>
struct ScnDftReal {
  typedef decltype(C{}.x) type;
};

template <int R>
struct ScnDftSplit {
  static constexpr int A = (R % 4 == 0 && R > 4) ? 4 : (R % 2 == 0) ? 2 : (R % 3 == 0) ? 3 : 5;
  static constexpr int B = R / A;
  static_assert(A * B == R && (R == 1 || A > 1), "lengths with prime factors 2, 3 and 5 only");
};

// v * W_R^e, e a compile-time constant after unrolling: the trivial rotations cost no multiply
template <int R, class C>
SCN_DFT_FN C scn_dft_twiddle(C v, int e) {
  e %= R;
  if (e == 0) return v;
  if (4 * e == R) return C{v.y, -v.x};       // -i
  if (2 * e == R) return C{-v.x, -v.y};      // -1
  if (4 * e == 3 * R) return C{-v.y, v.x};   // +i
  typedef typename ScnDftReal<C>::type real_t;
  const real_t h = (real_t)0.70710678118654752440;
  if (8 * e == R) return C{(v.x + v.y) * h, (v.y - v.x) * h};        // (1 - i) / sqrt 2
  if (8 * e == 3 * R) return C{(v.y - v.x) * h, -(v.x + v.y) * h};   // (-1 - i) / sqrt 2
  if (8 * e == 5 * R) return C{-(v.x + v.y) * h, (v.x - v.y) * h};   // (-1 + i) / sqrt 2
  if (8 * e == 7 * R) return C{(v.x - v.y) * h, (v.x + v.y) * h};    // (1 + i) / sqrt 2
  const real_t wr = (real_t)__builtin_cos(6.283185307179586476925286766559 * e / R);
  const real_t wi = -(real_t)__builtin_sin(6.283185307179586476925286766559 * e / R);
  return C{v.x * wr - v.y * wi, v.x * wi + v.y * wr};
}

template <int R, class C>
struct ScnDft {
  static SCN_DFT_FN void run(C (&v)[R]) {
    constexpr int A = ScnDftSplit<R>::A, B = ScnDftSplit<R>::B;
    C y[R];
#pragma unroll
    for (int b = 0; b < B; b++) {
      C t[A];
#pragma unroll
      for (int a = 0; a < A; a++) t[a] = v[B * a + b];
      ScnDft<A, C>::run(t);
#pragma unroll
      for (int p = 0; p < A; p++) y[p * B + b] = scn_dft_twiddle<R>(t[p], b * p);
    }
#pragma unroll
    for (int p = 0; p < A; p++) {
      C u[B];
#pragma unroll
      for (int b = 0; b < B; b++) u[b] = y[p * B + b];
      ScnDft<B, C>::run(u);
#pragma unroll
      for (int q = 0; q < B; q++) v[p + A * q] = u[q];
    }
  }
};

template <class C>
struct ScnDft<1, C> {
  static SCN_DFT_FN void run(C (&)[1]) {}
};
template <class C>
struct ScnDft<2, C> {
  static SCN_DFT_FN void run(C (&v)[2]) {
    const C a = v[0], b = v[1];
    v[0] = C{a.x + b.x, a.y + b.y};
    v[1] = C{a.x - b.x, a.y - b.y};
  }
};
template <class C>
struct ScnDft<3, C> {
  static SCN_DFT_FN void run(C (&v)[3]) {
    typedef typename ScnDftReal<C>::type real_t;
    const real_t S = (real_t)0.86602540378443864676, HALF = (real_t)0.5;  // sin(2 pi / 3)
    const C s{v[1].x + v[2].x, v[1].y + v[2].y}, d{v[1].x - v[2].x, v[1].y - v[2].y};
    const C m{v[0].x - HALF * s.x, v[0].y - HALF * s.y};
    const C u{S * d.y, -S * d.x};  // -i S d
    v[0] = C{v[0].x + s.x, v[0].y + s.y};
    v[1] = C{m.x + u.x, m.y + u.y};
    v[2] = C{m.x - u.x, m.y - u.y};
  }
};
template <class C>
struct ScnDft<4, C> {
  static SCN_DFT_FN void run(C (&v)[4]) {  // W_4 = -i
    const C t0{v[0].x + v[2].x, v[0].y + v[2].y}, t1{v[0].x - v[2].x, v[0].y - v[2].y};
    const C t2{v[1].x + v[3].x, v[1].y + v[3].y}, t3{v[1].x - v[3].x, v[1].y - v[3].y};
    v[0] = C{t0.x + t2.x, t0.y + t2.y};
    v[2] = C{t0.x - t2.x, t0.y - t2.y};
    v[1] = C{t1.x + t3.y, t1.y - t3.x};  // t1 - i t3
    v[3] = C{t1.x - t3.y, t1.y + t3.x};  // t1 + i t3
  }
};
template <class C>
struct ScnDft<5, C> {
  static SCN_DFT_FN void run(C (&v)[5]) {
    typedef typename ScnDftReal<C>::type real_t;
    const real_t C1 = (real_t)0.30901699437494742410, C2 = (real_t)-0.80901699437494742410;  // cos(2 pi / 5), cos(4 pi / 5)
    const real_t S1 = (real_t)0.95105651629515357212, S2 = (real_t)0.58778525229247312917;   // sin(2 pi / 5), sin(4 pi / 5)
    const C t1{v[1].x + v[4].x, v[1].y + v[4].y}, t2{v[2].x + v[3].x, v[2].y + v[3].y};
    const C t3{v[1].x - v[4].x, v[1].y - v[4].y}, t4{v[2].x - v[3].x, v[2].y - v[3].y};
    const C m1{v[0].x + C1 * t1.x + C2 * t2.x, v[0].y + C1 * t1.y + C2 * t2.y};
    const C m2{v[0].x + C2 * t1.x + C1 * t2.x, v[0].y + C2 * t1.y + C1 * t2.y};
    const C n1{S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y};
    const C n2{S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y};
    v[0] = C{v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y};
    v[1] = C{m1.x + n1.y, m1.y - n1.x};  // m1 - i n1
    v[4] = C{m1.x - n1.y, m1.y + n1.x};  // m1 + i n1
    v[2] = C{m2.x + n2.y, m2.y - n2.x};  // m2 - i n2
    v[3] = C{m2.x - n2.y, m2.y + n2.x};  // m2 + i n2
  }
};

template <int R, class C>
SCN_DFT_FN void scn_dft(C (&v)[R]) {
  ScnDft<R, C>::run(v);
}
