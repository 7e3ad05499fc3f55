// Is r = v - float(bf16_rn(v)) computed by v_dot2c_f32_bf16 bit-identical to the
// shift/and/sub sequence?  (candidate for the wgrad split: 2 ops instead of 3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* a, float* b, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (2 * i + 1 >= n) return;
  f32x2 v = {in[2 * i], in[2 * i + 1]};
  bf16x2 hi = __builtin_convertvector(v, bf16x2);
  unsigned c0, c1;   // opaque: the compiler's inline constant for {-1, 0} is not what the HW reads
  asm volatile("v_mov_b32 %0, 0xbf80\n\tv_mov_b32 %1, 0xbf800000" : "=v"(c0), "=v"(c1));
  const bf16x2 m0 = __builtin_bit_cast(bf16x2, c0), m1 = __builtin_bit_cast(bf16x2, c1);
  a[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(hi, m0, v[0], false);
  a[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(hi, m1, v[1], false);
  f32x2 r = v - __builtin_convertvector(hi, f32x2);
  b[2 * i] = r[0];
  b[2 * i + 1] = r[1];
}
int main() {
  const int n = 1 << 24;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    float f;
    memcpy(&f, &u, 4);
    if (!(f == f) || f - f != 0.f) f = (float)(rand() % 1000) * 1e-3f;   // no inf/nan
    if (i % 3 == 0) f = ((rand() % 20001) - 10000) * 1e-4f;              // typical magnitudes
    h[i] = f;
  }
  float *d, *a, *b;
  hipMalloc(&d, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 512, 256>>>(d, a, b, n);
  std::vector<float> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; ++i)
    if (memcmp(&ha[i], &hb[i], 4)) {
      if (bad < 8) printf("diff v=%a dot=%a ref=%a\n", h[i], ha[i], hb[i]);
      ++bad;
    }
  printf("checked %d values, %ld differ\n", n, bad);
  return 0;
}
