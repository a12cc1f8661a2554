// vil_geom.h -- chunk / neighbour / mask / bias-index geometry shared by every
// kernel family and by the host-side helpers (vil_geom_mask, vil_geom_bias_index)
// that the CPU tests pin against the golden masks.
//
// Restates, in closed form, the reference's index conventions (paths relative to
// the reference repository):
//   chunking + bottom/right zero padding   src/models/layers/longformer2d.py:134-149
//   neighbour order / random-shift modes   src/models/layers/slidingchunk_2d.py:15-24,37-79
//   zero / cyclic / exact masks            src/models/layers/slidingchunk_2d.py:249-318
//   relative position index                src/models/layers/longformer2d.py:67-100
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VIL_HD __host__ __device__ __forceinline__
#else
#define VIL_HD inline
#endif

enum { VIL_KEY_MASKED = 0, VIL_KEY_REAL = 1, VIL_KEY_ZERO = 2 };

struct VilGeom {
  int nx, ny;        // real local grid
  int W, W2;         // chunk side, tokens per chunk
  int mx, my;        // chunks per side after padding
  int exact, mode;
  int nact;          // active neighbour offsets: 9 / 1 / 2
  int tbl;           // 4W-1 (side of the bias table)
  int adr[9], adc[9];
};

VIL_HD void vil_geom_init(VilGeom& g, int nx, int ny, int W, int exact, int mode) {
  g.nx = nx; g.ny = ny; g.W = W; g.W2 = W * W;
  g.mx = (nx + W - 1) / W; g.my = (ny + W - 1) / W;
  g.exact = exact; g.mode = mode; g.tbl = 4 * W - 1;
  if (mode == 0) {
    g.nact = 9;
    for (int s = 0; s < 9; ++s) { g.adr[s] = s / 3 - 1; g.adc[s] = s % 3 - 1; }
  } else if (mode < 0) {
    g.nact = 1; g.adr[0] = 0; g.adc[0] = 0;
  } else {
    // random-shift mode m: [own chunk, neighbour m]; m<=4 -> slot m-1, m>4 -> slot m
    const int s = mode > 4 ? mode : mode - 1;
    g.nact = 2; g.adr[0] = 0; g.adc[0] = 0; g.adr[1] = s / 3 - 1; g.adc[1] = s % 3 - 1;
  }
  for (int s = g.nact; s < 9; ++s) { g.adr[s] = 0; g.adc[s] = 0; }
}

// positive modulo
VIL_HD int vil_pmod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }

// State of the key at in-chunk position (xt,yt) of the chunk at offset (dr,dc)
// from query chunk (m,n).  REAL -> tok = local token index kr*ny+kc;
// ZERO (cyclic mode only) -> a zero-padded position that stays in the softmax;
// MASKED -> not attended.  Query-position independent (exact==1 adds
// vil_exact_window on top).
VIL_HD int vil_key_state(const VilGeom& g, int m, int n, int dr, int dc, int xt, int yt,
                         int& kr, int& kc) {
  if (g.exact == -1) {
    const bool masked = ((m + dr + 1 == g.mx) && ((g.mx - 1) * g.W + xt >= g.nx)) ||
                        ((n + dc + 1 == g.my) && ((g.my - 1) * g.W + yt >= g.ny));
    kr = vil_pmod(m + dr, g.mx) * g.W + xt;
    kc = vil_pmod(n + dc, g.my) * g.W + yt;
    if (masked) return VIL_KEY_MASKED;
    if (kr >= g.nx || kc >= g.ny) return VIL_KEY_ZERO;
    return VIL_KEY_REAL;
  }
  const int cm = m + dr, cn = n + dc;
  kr = cm * g.W + xt; kc = cn * g.W + yt;
  if (cm < 0 || cm >= g.mx || cn < 0 || cn >= g.my) return VIL_KEY_MASKED;
  if (kr >= g.nx || kc >= g.ny) return VIL_KEY_MASKED;
  return VIL_KEY_REAL;
}

// exact==1: key (kr,kc) is inside the (2W+1)^2 window of query (qr,qc)
VIL_HD bool vil_exact_window(int W, int qr, int qc, int kr, int kc) {
  const int a = kr - qr, b = kc - qc;
  return a <= W && a >= -W && b <= W && b >= -W;
}

// index into the ((4W-1)^2) relative-position-bias table for query (xl,yl) of
// the centre chunk and key (xt,yt) of the chunk at offset (dr,dc)
VIL_HD int vil_bias_index(int W, int xl, int yl, int dr, int dc, int xt, int yt) {
  return (xl - (dr * W + xt) + 2 * W - 1) * (4 * W - 1) + (yl - (dc * W + yt) + 2 * W - 1);
}
