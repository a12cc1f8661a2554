// Probe of LDS-DMA through a bounded buffer descriptor (`buffer_load_dwordx4 ... lds`, __builtin_amdgcn_raw_ptr_buffer_load_lds):
// LDS destination = wave-uniform base + lane * 16; the XOR swizzle of the dense-stage kernels is applied to the SOURCE chunk;
// out-of-range dwords arrive as zeros (range check per dword).   hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const char* g, char* out, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, (short)0, n, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane -> row lane/8, slot lane%8 holds logical chunk (slot ^ f(row))
  const int row = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ (((row >> 1) & 3) << 1);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, (wave * 8 + row) * 128 + chunk * 16, 0, 0, 0);
  __syncthreads();
  ((u32x4*)out)[threadIdx.x] = ((u32x4*)smem)[threadIdx.x];
}
int main() {
  const int n = 4096;
  char *g, *o; hipMalloc(&g, n); hipMalloc(&o, n);
  unsigned short h[2048]; for (int i = 0; i < 2048; ++i) h[i] = i;
  hipMemcpy(g, h, n, hipMemcpyHostToDevice);
  k<<<1, 256, 4096>>>(g, o, 3000);   // bound = 3000 bytes: the chunk that straddles it keeps its in-range dwords, everything beyond is zero
  unsigned short r[2048]; hipMemcpy(r, o, n, hipMemcpyDeviceToHost);
  // expect LDS row R slot S holds logical chunk S ^ f(R): values chunk*8..+7 + R*64 ; zeros beyond 3000 bytes
  int bad = 0;
  for (int R = 0; R < 32; ++R) for (int S = 0; S < 8; ++S) for (int e = 0; e < 8; ++e) {
    int chunk = S ^ (((R >> 1) & 3) << 1);
    int src = R * 64 + chunk * 8 + e;
    int want = (R * 128 + chunk * 16 + (e / 2) * 4 + 4 <= 3000) ? src : 0;       // per-dword range check
    if (r[R * 64 + S * 8 + e] != want) { if (bad < 10) printf("R%d S%d e%d got %d want %d\n", R, S, e, r[R*64+S*8+e], want); ++bad; }
  }
  printf(bad ? "FAIL %d\n" : "PASS\n", bad);
  return bad != 0;
}
