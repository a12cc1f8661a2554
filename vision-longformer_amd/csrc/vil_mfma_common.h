// vil_mfma_common.h -- types, launch configuration and LDS bias-table layout shared by the
// MFMA forward and backward kernels.
#pragma once
#include "vil_internal.h"
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define VIL_CPAD 3   // left pad (floats) of every bias-table row in LDS
#define VIL_MASK_VAL (-1.0e30f)
#define VIL_M_INIT (-1.0e20f)
#define LOG2E 1.4426950408889634f

struct MfmaCfg {
  int P;             // row pitch (floats) of the LDS bias table
  int copysize;      // floats per table copy (multiple of 4)
  int cstride_b;     // (copysize - 1) * 4: byte offset between consecutive shifted copies
  int guard0;        // start (floats) of the all-masked region
  int glo0;          // start of the per-global-token constant regions
  int gsz;           // size of one such region (Aq range + 4)
  int aconst;        // (2W-1)*(P+1) + VIL_CPAD
  unsigned magicW, magicW2;
  int HQ;            // query quads per chunk row = ceil(W/4)
  int NWP;           // waves per chunk = ceil(W*HQ/16)
  int NS;            // real key slots = G + nact*W2
  int NSP;           // padded to a multiple of 32
  int units_bh;      // mx*my*NWP
  int wg_per_bh, gpw;
  int wpw;           // waves per workgroup (4, or fewer when the per-wave LDS is large)
  int wave_lds;      // bytes of private LDS per wave
  int no_tr;         // debug: read V^T with scalar LDS loads instead of ds_read_b64_tr_b16
  const float* tabws;  // (H, 4*copysize) prepared bias tables
};

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned magic) { return __umulhi(n, magic); }


// host: fills the launch configuration for a descriptor
bool vil_mfma_make_cfg(const VilAttnDesc* d, MfmaCfg& c);
// device prologue kernel: builds the 4 shifted copies of every head's bias table
__global__ void k_mfma_table(VilParams p, MfmaCfg c, float* out);
