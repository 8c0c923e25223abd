// learner_problems.h — the operand gathers / epilogues ("problems") that bind gemm_core.h's
// MFMA skeleton to each contraction of the Rainbow learn step (model.py:42-46,55-80 forward;
// their autograd adjoints for agent.py:96).
//
// Image ordering of a learn step (NI = 3B activations rows / images):
//   [0,B)   online net on states        (the only pass with a backward)   agent.py:66
//   [B,2B)  online net on next_states   (double-Q action selection)       agent.py:71
//   [2B,3B) target net on next_states                                     agent.py:75
// "net 0" = online (images [0,2B)), "net 1" = target (images [2B,3B)).
//
// Activations are NCHW float32, conv weights OIHW, linear weights [out][in] exactly like the
// reference state dict, so checkpoints interchange without re-layout.
#pragma once
#include "gemm_core.h"

// exact u8 -> x/255 (memory.py:137-138 `.div_(255)`) without a divide: one multiply by fl(1/255) plus one
// Newton correction is the correctly rounded quotient for all 256 byte values (checked exhaustively in
// tests/test_learner_emu.py::test_unit_conversion_is_exact).
__device__ __forceinline__ float rb_unit(uint8_t u) {
  const float x = (float)u;
  const float inv = 1.0f / 255.0f;
  const float q = x * inv;
  const float r = fmaf(-255.0f, q, x);
  return fmaf(r, inv, q);
}

template <int KS_, int S_, int IH_, int OH_>
struct ConvGeom {
  static constexpr int KS = KS_, S = S_, IH = IH_, OH = OH_;
  static constexpr int P = OH_ * OH_, KK = KS_ * KS_, IP = IH_ * IH_;
};

// Where the first conv layer reads its images from.
struct ImgSrc {
  const uint8_t* u8_states;       // [B][cin][84*84]   images [0,B)
  const uint8_t* u8_next;         // [B][cin][84*84]   images [B,2B) and [2B,3B)
  const float* f32;               // act(): [n][cin][84*84] already in [0,1]
  int B;
  // zero-copy mode: frames are read straight from the replay ring through the sampler's window table
  // (memory.py:112-121: slot c of the state stack, slot n+c of the next-state stack, -1 = blanked frame)
  const uint8_t* ring;            // [capacity][84*84] or NULL
  const int32_t* win;             // [B][win_len]
  int win_len, n_step;
};

// frame (img, channel c) as a u8 pointer, or nullptr for a blanked frame
__device__ __forceinline__ const uint8_t* rb_frame_ptr(const ImgSrc& s, int img, int c, int cin, int ip) {
  if (s.ring) {
    const int sample = img < s.B ? img : (img - s.B) % s.B;
    const int slot = img < s.B ? c : s.n_step + c;
    const int32_t idx = s.win[(int64_t)sample * s.win_len + slot];
    return idx < 0 ? nullptr : s.ring + (int64_t)idx * ip;
  }
  const uint8_t* base = img < s.B ? s.u8_states + (int64_t)img * cin * ip : s.u8_next + (int64_t)((img - s.B) % s.B) * cin * ip;
  return base + (int64_t)c * ip;
}

struct NetPtrs {
  const float* conv_w[3];
  const float* conv_b[3];
  const float *h_mu, *h_sigma, *h_bmu, *h_bsigma;   // [2H][F], [2H]   rows: value stream then advantage stream
  const float *z_mu, *z_sigma, *z_bmu, *z_bsigma;   // [NZ][H], [NZ]   rows: Z value atoms then A*Z advantage atoms
  const float *h_ein, *h_eout, *z_ein, *z_eout;     // f(eps): [2][F], [2H], [2][H], [NZ]   (model.py:36-40)
};

// ------------------------------------------------------------------ conv forward --
// out[img][co][pos] = relu(bias[co] + sum_k W[co][k] * im2col[k][pos])      model.py:56-58,61-62
template <class G, bool U8IN>
struct ConvFwdProb {
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int cin, cout;
  int n_img[2], img_base[2];
  const float* w[2];
  const float* bias[2];
  ImgSrc src;          // U8IN
  const float* in_f;   // !U8IN: [img][cin][IP]
  float* out;          // [img][cout][P]

  __device__ bool group(int g, GemmDims& d) const {
    if (n_img[g] == 0) return false;
    d.M = cout; d.N = n_img[g] * G::P; d.K = cin * G::KK; d.k_begin = 0; d.k_end = d.K;
    return true;
  }
  __device__ float a(int g, int m, int k) const { return w[g][m * (cin * G::KK) + k]; }
  __device__ float b(int g, int k, int n) const {
    const int img = img_base[g] + n / G::P;
    const int rem = n % G::P;
    const int oy = rem / G::OH, ox = rem % G::OH;
    const int c = k / G::KK, r = k % G::KK;
    const int ky = r / G::KS, kx = r % G::KS;
    const int idx = (c * G::IH + oy * G::S + ky) * G::IH + ox * G::S + kx;
    if (U8IN) {
      if (src.f32) return src.f32[(int64_t)img * cin * G::IP + idx];
      const uint8_t* base = img < src.B ? src.u8_states + (int64_t)img * cin * G::IP
                                        : src.u8_next + (int64_t)((img - src.B) % src.B) * cin * G::IP;
      return rb_unit(base[idx]);
    }
    return in_f[(int64_t)img * cin * G::IP + idx];
  }
  __device__ void store(int g, int m, int n, float v) const {
    const int img = img_base[g] + n / G::P;
    const int rem = n % G::P;
    out[((int64_t)img * cout + m) * G::P + rem] = fmaxf(v + bias[g][m], 0.0f);
  }
};

// --------------------------------------------------------- noisy linear forward --
// W[n][k] = mu + sigma * (eps_out[n] * eps_in[k]) formed in registers with the reference's
// rounding order (model.py:39,44); never written to memory.
__device__ __forceinline__ float rb_noisy_w(const float* mu, const float* sigma, const float* eout, const float* ein,
                                            int64_t row, int64_t ld, int k, int ein_off) {
  const float eps_w = eout[row] * ein[ein_off + k];
  return mu[row * ld + k] + sigma[row * ld + k] * eps_w;
}

// hidden layer, both streams at once: part[s][img][n] = sum_{k in split s} feat[img][k] * W[n][k]
struct FcHFwdProb {
  static constexpr bool A_KFAST = true, B_KFAST = true;
  int F, H, NI, splits;
  int n_img[2], img_base[2];
  const float* feat;
  NetPtrs net[2];
  float* part;   // [splits][NI][2H]

  __device__ bool group(int g, GemmDims& d) const {
    const int nt = g & 1, s = g >> 1;
    if (n_img[nt] == 0) return false;
    d.M = n_img[nt]; d.N = 2 * H; d.K = F;
    rb_split_k(F, splits, s, &d.k_begin, &d.k_end);
    return d.k_end > d.k_begin;
  }
  __device__ float a(int g, int m, int k) const { return feat[(int64_t)(img_base[g & 1] + m) * F + k]; }
  __device__ float b(int g, int k, int n) const {
    const NetPtrs& p = net[g & 1];
    return rb_noisy_w(p.h_mu, p.h_sigma, p.h_eout, p.h_ein, n, F, k, n >= H ? F : 0);
  }
  __device__ void store(int g, int m, int n, float v) const {
    part[((int64_t)(g >> 1) * NI + img_base[g & 1] + m) * (2 * H) + n] = v;
  }
};

// output layer: logits[img][n0+n] = b[n0+n] + sum_k h[img][stream*H+k] * W[n0+n][k]
struct FcZFwdProb {
  static constexpr bool A_KFAST = true, B_KFAST = true;
  int H, Z, NZ;
  int n_img[2], img_base[2];
  const float* h;   // [NI][2H]
  NetPtrs net[2];
  float* logits;    // [NI][NZ]

  __device__ bool group(int g, GemmDims& d) const {
    const int nt = g & 1, st = g >> 1;
    if (n_img[nt] == 0) return false;
    d.M = n_img[nt]; d.N = st ? NZ - Z : Z; d.K = H; d.k_begin = 0; d.k_end = H;
    return true;
  }
  __device__ float a(int g, int m, int k) const {
    return h[(int64_t)(img_base[g & 1] + m) * (2 * H) + (g >> 1) * H + k];
  }
  __device__ float b(int g, int k, int n) const {
    const NetPtrs& p = net[g & 1];
    const int st = g >> 1;
    return rb_noisy_w(p.z_mu, p.z_sigma, p.z_eout, p.z_ein, (st ? Z : 0) + n, H, k, st * H);
  }
  __device__ void store(int g, int m, int n, float v) const {
    const NetPtrs& p = net[g & 1];
    const int row = ((g >> 1) ? Z : 0) + n;
    const float bias = p.z_bmu[row] + p.z_bsigma[row] * p.z_eout[row];      // model.py:44
    logits[(int64_t)(img_base[g & 1] + m) * NZ + row] = v + bias;
  }
};

// ------------------------------------------------------ noisy linear backward --
// d mu = dY^T X ; d sigma = d mu * eps_w ; bias grads ride along as the extra column k == K
// (X == 1).  One writer per gradient element: deterministic.
struct FcGradOut {
  float *g_mu, *g_sigma, *g_bmu, *g_bsigma;
  const float *eout, *ein;
};
__device__ __forceinline__ void rb_store_noisy_grad(const FcGradOut& o, int64_t row, int in_f, int k, int ein_off,
                                                    float v) {
  if (k < in_f) {
    o.g_mu[row * in_f + k] = v;
    o.g_sigma[row * in_f + k] = v * (o.eout[row] * o.ein[ein_off + k]);
  } else {
    o.g_bmu[row] = v;
    o.g_bsigma[row] = v * o.eout[row];
  }
}

// fc_z weight grads: group = stream
struct FcZDwProb {
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int B, H, Z, NZ;
  const float* dlogits;   // [B][NZ]
  const float* h;         // [NI][2H], rows [0,B)
  FcGradOut o;
  __device__ bool group(int g, GemmDims& d) const {
    d.M = g ? NZ - Z : Z; d.N = H + 1; d.K = B; d.k_begin = 0; d.k_end = B;
    return true;
  }
  __device__ float a(int g, int m, int k) const { return dlogits[(int64_t)k * NZ + (g ? Z : 0) + m]; }
  __device__ float b(int g, int k, int n) const { return n == H ? 1.0f : h[(int64_t)k * (2 * H) + g * H + n]; }
  __device__ void store(int g, int m, int n, float v) const {
    rb_store_noisy_grad(o, (g ? Z : 0) + m, H, n, g * H, v);
  }
};

// fc_z input grads (+ ReLU mask of the hidden layer): dh[b][stream*H+k]
struct FcZDxProb {
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int B, H, Z, NZ;
  const float* dlogits;
  const float* h;
  NetPtrs net;
  float* dh;   // [B][2H]
  __device__ bool group(int g, GemmDims& d) const {
    d.M = B; d.N = H; d.K = g ? NZ - Z : Z; d.k_begin = 0; d.k_end = d.K;
    return true;
  }
  __device__ float a(int g, int m, int k) const { return dlogits[(int64_t)m * NZ + (g ? Z : 0) + k]; }
  __device__ float b(int g, int k, int n) const {
    return rb_noisy_w(net.z_mu, net.z_sigma, net.z_eout, net.z_ein, (g ? Z : 0) + k, H, n, g * H);
  }
  __device__ void store(int g, int m, int n, float v) const {
    const int64_t i = (int64_t)m * (2 * H) + g * H + n;
    dh[i] = h[i] > 0.0f ? v : 0.0f;
  }
};

// fc_h weight grads: rows n of [2H], columns k of [F] (+ bias column)
struct FcHDwProb {
  static constexpr bool A_KFAST = false, B_KFAST = false;
  int B, H, F;
  const float* dh;     // [B][2H]
  const float* feat;   // [NI][F], rows [0,B)
  FcGradOut o;
  __device__ bool group(int, GemmDims& d) const {
    d.M = 2 * H; d.N = F + 1; d.K = B; d.k_begin = 0; d.k_end = B;
    return true;
  }
  __device__ float a(int, int m, int k) const { return dh[(int64_t)k * (2 * H) + m]; }
  __device__ float b(int, int k, int n) const { return n == F ? 1.0f : feat[(int64_t)k * F + n]; }
  __device__ void store(int, int m, int n, float v) const { rb_store_noisy_grad(o, m, F, n, m >= H ? F : 0, v); }
};

// fc_h input grads, split over the 2H reduction: part[s][b][k]
struct FcHDxProb {
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int B, H, F, splits;
  const float* dh;
  NetPtrs net;
  float* part;   // [splits][B][F]
  __device__ bool group(int g, GemmDims& d) const {
    d.M = B; d.N = F; d.K = 2 * H;
    rb_split_k(2 * H, splits, g, &d.k_begin, &d.k_end);
    return d.k_end > d.k_begin;
  }
  __device__ float a(int, int m, int k) const { return dh[(int64_t)m * (2 * H) + k]; }
  __device__ float b(int, int k, int n) const {
    return rb_noisy_w(net.h_mu, net.h_sigma, net.h_eout, net.h_ein, k, F, n, k >= H ? F : 0);
  }
  __device__ void store(int g, int m, int n, float v) const { part[((int64_t)g * B + m) * F + n] = v; }
};

// ---------------------------------------------------------------- conv backward --
// weight grads: part[s][co][col] = sum_{pos in split s} dY[co][pos] * im2col[pos][col],
// col == K is the bias column.
template <class G, bool U8IN>
struct ConvDwProb {
  static constexpr bool A_KFAST = true, B_KFAST = true;
  int B, cin, cout, splits;
  const float* dy;       // [B][cout][P]
  const uint8_t* x_u8;   // U8IN: states [B][cin][IP]
  const float* x_f;      // else: previous activation [NI][cin][IP], rows [0,B)
  float* part;           // [splits][cout][K+1]
  __device__ bool group(int g, GemmDims& d) const {
    d.M = cout; d.N = cin * G::KK + 1; d.K = B * G::P;
    rb_split_k(d.K, splits, g, &d.k_begin, &d.k_end);
    return d.k_end > d.k_begin;
  }
  __device__ float a(int, int m, int k) const {
    const int img = k / G::P, rem = k % G::P;
    return dy[((int64_t)img * cout + m) * G::P + rem];
  }
  __device__ float b(int, int k, int n) const {
    if (n == cin * G::KK) return 1.0f;
    const int img = k / G::P, rem = k % G::P;
    const int oy = rem / G::OH, ox = rem % G::OH;
    const int c = n / G::KK, r = n % G::KK;
    const int ky = r / G::KS, kx = r % G::KS;
    const int64_t idx = (int64_t)img * cin * G::IP + (c * G::IH + oy * G::S + ky) * G::IH + ox * G::S + kx;
    return U8IN ? rb_unit(x_u8[idx]) : x_f[idx];
  }
  __device__ void store(int g, int m, int n, float v) const {
    part[((int64_t)g * cout + m) * (cin * G::KK + 1) + n] = v;
  }
};

// data grads through a strided conv, decomposed by output phase (y % S, x % S) so that every
// staged tap is a real tap; epilogue applies the ReLU mask of the producing layer.
// group = phase.  C[c][(img,yy,xx)] = sum_{co,ty,tx} W[co][c][py+ty*S][px+tx*S] * dY[img][co][yy-ty][xx-tx]
template <class G>
struct ConvDxProb {
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int B, cin, cout;
  const float* w;        // [cout][cin][KS][KS]
  const float* dy;       // [B][cout][P]
  const float* x_act;    // input activation of this layer (post-ReLU) [NI][cin][IP], rows [0,B)
  float* dx;             // [B][cin][IP]
  struct Phase { int py, px, nty, ntx, nyy, nxx; };
  __device__ Phase phase(int g) const {
    Phase f;
    f.py = g / G::S; f.px = g % G::S;
    f.nty = (G::KS - f.py + G::S - 1) / G::S; f.ntx = (G::KS - f.px + G::S - 1) / G::S;
    f.nyy = (G::IH - f.py + G::S - 1) / G::S; f.nxx = (G::IH - f.px + G::S - 1) / G::S;
    return f;
  }
  __device__ bool group(int g, GemmDims& d) const {
    const Phase f = phase(g);
    d.M = cin; d.N = B * f.nyy * f.nxx; d.K = cout * f.nty * f.ntx; d.k_begin = 0; d.k_end = d.K;
    return d.K > 0 && d.N > 0;
  }
  __device__ float a(int g, int m, int k) const {
    const Phase f = phase(g);
    const int taps = f.nty * f.ntx;
    const int co = k / taps, r = k % taps;
    const int ky = f.py + (r / f.ntx) * G::S, kx = f.px + (r % f.ntx) * G::S;
    return w[(((int64_t)co * cin + m) * G::KS + ky) * G::KS + kx];
  }
  __device__ float b(int g, int k, int n) const {
    const Phase f = phase(g);
    const int taps = f.nty * f.ntx;
    const int co = k / taps, r = k % taps;
    const int ty = r / f.ntx, tx = r % f.ntx;
    const int per = f.nyy * f.nxx;
    const int img = n / per, q = n % per;
    const int oy = q / f.nxx - ty, ox = q % f.nxx - tx;
    if (oy < 0 || ox < 0 || oy >= G::OH || ox >= G::OH) return 0.0f;
    return dy[((int64_t)img * cout + co) * G::P + oy * G::OH + ox];
  }
  __device__ void store(int g, int m, int n, float v) const {
    const Phase f = phase(g);
    const int per = f.nyy * f.nxx;
    const int img = n / per, q = n % per;
    const int y = (q / f.nxx) * G::S + f.py, x = (q % f.nxx) * G::S + f.px;
    const int64_t idx = ((int64_t)img * cin + m) * G::IP + y * G::IH + x;
    dx[idx] = x_act[idx] > 0.0f ? v : 0.0f;
  }
};
