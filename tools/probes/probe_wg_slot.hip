// probe: where do the workgroups of a 3-per-CU persistent grid land?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o probe_wg_slot probe_wg_slot.hip && ./probe_wg_slot
// 768 workgroups x 256 threads x 50 048 B of dynamic LDS (the i8 depth kernel's shape).  Every workgroup records HW_REG_HW_ID (CU, SE, wave slot),
// HW_REG_XCC_ID and HW_REG_LDS_ALLOC (LDS base / size) and then spins so that the whole grid is resident at once.
// Question: is "slot on the CU" = blockIdx / 256, and is the LDS base a usable slot index?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 3) void probe(unsigned* out, int spin)
{
  extern __shared__ unsigned char lds[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const unsigned la = __builtin_amdgcn_s_getreg((31 << 11) | 6);    // HW_REG_LDS_ALLOC
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = hw;
    out[4 * blockIdx.x + 1] = la;
    out[4 * blockIdx.x + 2] = xcc;
    out[4 * blockIdx.x + 3] = (unsigned)(t0 & 0xFFFFFFFFu);
  }
}

int main()
{
  const int G = 768;
  unsigned* d;
  hipMalloc(&d, G * 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 50048);
  probe<<<G, 256, 50048>>>(d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(G * 4);
  hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> per_cu;
  int agree = 0;
  for (int b = 0; b < G; ++b) {
    const unsigned hw = h[4 * b], la = h[4 * b + 1], xcc = h[4 * b + 2] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;  // gfx9 layout: CU_ID[11:8], SH_ID[12], SE_ID[15:13]
    const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
    per_cu[key].push_back(b);
    if (b < 24 || b % 97 == 0) printf("block %3d: xcc %u se %u sh %u cu %2u  hw_id %08x  lds_alloc %08x (base field %u, size field %u)\n", b, xcc, se, sh, cu, hw, la, la & 0xFF, (la >> 12) & 0x1FF);
  }
  printf("distinct CUs: %zu\n", per_cu.size());
  std::map<size_t, int> hist;
  for (auto& kv : per_cu) {
    hist[kv.second.size()]++;
    std::vector<int> slots;
    for (int b : kv.second) slots.push_back(b / 256);
    bool distinct = true;
    for (size_t i = 0; i < slots.size(); ++i) for (size_t j = i + 1; j < slots.size(); ++j) if (slots[i] == slots[j]) distinct = false;
    if (distinct && kv.second.size() == 3) ++agree;
  }
  for (auto& kv : hist) printf("CUs holding %zu workgroups: %d\n", kv.first, kv.second);
  printf("CUs whose three workgroups have three different blockIdx / 256: %d\n", agree);
  // LDS base per block vs blockIdx / 256
  std::map<unsigned, std::map<int, int>> base_vs_slot;
  for (int b = 0; b < G; ++b) base_vs_slot[h[4 * b + 1] & 0xFF][b / 256]++;
  for (auto& kv : base_vs_slot) { printf("lds base field %3u:", kv.first); for (auto& s : kv.second) printf("  slot %d x %d", s.first, s.second); printf("\n"); }
  return 0;
}
