#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4;
__global__ void k(unsigned short* out, const int* addr_of_lane) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr_of_lane[threadIdx.x];   // element index (multiple of 4)
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int h_addr[64];
  // lane l = 16 g + i supplies row (i>>2) of its group's block, column quad (i&3);
  // rows are 40 elements apart (stride free?), groups 256 elements apart
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h_addr[l] = g * 256 + (i >> 2) * 40 + (i & 3) * 4; }
  int* d_addr; unsigned short* d_out; unsigned short h_out[256];
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int g = l >> 4, i = l & 15;
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      int expect = g * 256 + j * 40 + i;     // (row j, col i) of the group's block
      printf(" %4d%s", h_out[l * 4 + j], h_out[l * 4 + j] == expect ? "" : "!");
      bad += h_out[l * 4 + j] != expect;
    }
    printf("\n");
  }
  printf("MISMATCHES %d\n", bad);
  return 0;
}
