#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned long long* out){
  __shared__ __attribute__((aligned(16))) unsigned short tile[64*64];
  for (int i=threadIdx.x;i<64*64;i+=64) tile[i]=in[i];
  __syncthreads();
  const int lane=threadIdx.x, g=lane/16, q=(lane%16)/4, p=lane%4;
  auto ptr = (__attribute__((address_space(3))) v4s*)(tile + (4*g+q)*64 + 4*p);
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  out[lane] = __builtin_bit_cast(unsigned long long, v);
}
int main(){
  std::vector<unsigned short> h(64*64); for(int r=0;r<64;r++)for(int c=0;c<64;c++)h[r*64+c]=r*64+c;
  unsigned short* d; unsigned long long* o; hipMalloc(&d,h.size()*2); hipMalloc(&o,64*8);
  hipMemcpy(d,h.data(),h.size()*2,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,d,o);
  unsigned long long r[64]; hipMemcpy(r,o,sizeof(r),hipMemcpyDeviceToHost);
  int bad=0;
  for(int l=0;l<64;l++){ int g=l/16,i=l%16; for(int q=0;q<4;q++){ unsigned v=(r[l]>>(16*q))&0xffff; unsigned e=(4*g+q)*64+i; if(v!=e){bad++; if(bad<8) printf("lane %d q %d got %u (row %u col %u) expected %u\n",l,q,v,v/64,v%64,e);} } }
  printf("tr_read check: %s (%d mismatches)\n", bad?"MISMATCH":"OK", bad); return 0; }
