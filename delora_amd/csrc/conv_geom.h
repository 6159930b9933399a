// Geometry policies of the convolution kernels (conv.hip: fp32, convh.hip: bf16 / fp16), shared by both families.
#pragma once

// Geometry policies: which offsets of the staged input tile feed which weight tap, and where an output pixel goes.
//   GeomConv<KS, SH, SW>            the layer itself: output (ho, wo) reads input (ho*SH + r - PAD, wo*SW + s - PAD).
//   GeomDgrad<KS, SH, SW, PH, PW, SO> its input gradient for the input pixels (SH*i + PH, SW*j + PW): a stride-1 pass over
//       the OUTPUT-gradient grid with the subset of taps whose stride phase matches -- r with (PH + PAD - r) % SH == 0
//       reads grid row i + (PH + PAD - r) / SH -- so a strided layer's input gradient is SH*SW such passes, each doing
//       exactly the multiplications that are not zeros (no zero-stuffed tensor).  SO: scatter the result into the
//       full-resolution image (stride SH, SW, phase PH, PW) or keep it dense on the grid.
template <int KS_, int SH_, int SW_>
struct GeomConv {
  static constexpr int NT = KS_ * KS_, WTAPS = KS_ * KS_, ISH = SH_, ISW = SW_;
  static constexpr int H0 = -((KS_ - 1) / 2), W0 = -((KS_ - 1) / 2), EH = KS_, EW = KS_;
  static constexpr int OSH = 1, OSW = 1, OPH = 0, OPW = 0;
  static constexpr int dh(int t) { return t / KS_; }
  static constexpr int dw(int t) { return t % KS_; }
  static constexpr int wt(int t) { return t; }
};

template <int KS_, int S_, int P_>
struct DgradAxis {                                 // one axis of GeomDgrad: valid taps and their grid offsets
  static constexpr int PAD = (KS_ - 1) / 2;
  static constexpr bool valid(int r) { return (P_ + PAD - r) % S_ == 0; }
  static constexpr int off(int r) { return (P_ + PAD - r) / S_; }
  static constexpr int count() { int n = 0; for (int r = 0; r < KS_; ++r) n += valid(r) ? 1 : 0; return n; }
  static constexpr int tap(int i) { int n = 0; for (int r = 0; r < KS_; ++r) if (valid(r)) { if (n == i) return r; ++n; } return 0; }
  static constexpr int lo() { int m = 99; for (int r = 0; r < KS_; ++r) if (valid(r) && off(r) < m) m = off(r); return m; }
  static constexpr int hi() { int m = -99; for (int r = 0; r < KS_; ++r) if (valid(r) && off(r) > m) m = off(r); return m; }
};

template <int KS_, int SH_, int SW_, int PH_, int PW_, bool SO_>
struct GeomDgrad {
  using AH = DgradAxis<KS_, SH_, PH_>;
  using AW = DgradAxis<KS_, SW_, PW_>;
  static constexpr int NR = AH::count(), NS = AW::count();
  static constexpr int NT = NR * NS, WTAPS = KS_ * KS_, ISH = 1, ISW = 1;
  static constexpr int H0 = AH::lo(), W0 = AW::lo(), EH = AH::hi() - AH::lo() + 1, EW = AW::hi() - AW::lo() + 1;
  static constexpr int OSH = SO_ ? SH_ : 1, OSW = SO_ ? SW_ : 1, OPH = SO_ ? PH_ : 0, OPW = SO_ ? PW_ : 0;
  static constexpr int dh(int t) { return AH::off(AH::tap(t / NS)) - H0; }
  static constexpr int dw(int t) { return AW::off(AW::tap(t % NS)) - W0; }
  static constexpr int wt(int t) { return AH::tap(t / NS) * KS_ + AW::tap(t % NS); }
};

