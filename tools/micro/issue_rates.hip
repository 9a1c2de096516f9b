// Microbenchmark (tools/ only, not the product): issue cost of the instructions the scan kernels are
// made of, on gfx950, at 1, 2 and 4 waves per SIMD (the scan kernels run one 1,024-thread block per CU).
// Per test: every wave runs ITER iterations of 64 instructions (8 independent chains), stamped with
// s_memtime; printed = shader cycles per instruction per WAVE and per SIMD (wave cycles / waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

constexpr int ITER = 2000;

template <int T>
__global__ __launch_bounds__(1024) void k_issue(unsigned long long *out, unsigned *sink, unsigned seed) {
  extern __shared__ unsigned lds[];
  for (unsigned i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = i * 2654435761u;
  __syncthreads();
  unsigned a[8];
  unsigned lane = threadIdx.x & 63;
  for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x * 7 + i;
  unsigned b = seed | 1, c = seed * 3 + 5;
  unsigned long long m = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b));
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(c));
#define OP_PKMAD(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define OP_PKMIN(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PKSUB(i) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(c));
#define OP_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_SDWASUB(i) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
#define OP_SDWASHL(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(b));
#define OP_BFE(i) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(b));
#define OP_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_ADDLSHL(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
#define OP_MBCNT(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"(c));
#define OP_CMP(i) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define OP_CMPADDC(i) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define OP_SALU(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(c) : "s"(b));
#define OP_SBCNT(i) asm volatile("s_bcnt1_i32_b64 %0, vcc\n s_add_u32 %1, %1, %0" : "=s"(b), "+s"(c) : : "scc");
    // v_cmp -> s_cbranch_vccz chain (never taken: vcc nonzero), the per-position tail of the gram kernel
#define OP_CMPBR(i) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %1\n1:" : "+v"(a[i]) : "v"(b) : "vcc");
    // v_cmp -> saveexec -> valu -> restore (the push under exec)
#define OP_SAVEEXEC(i) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n s_and_saveexec_b64 s[30:31], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[30:31]" : "+v"(a[i]) : "v"(b) : "vcc", "s30", "s31");
    // LDS reads: u16 at random addresses (bank conflicts), at lane-private banks, b32 random, d16_hi
#define OP_LDSU16(i) { unsigned r; asm volatile("ds_read_u16 %0, %1" : "=v"(r) : "v"(a[i] & 0x1fffeu)); a[i] += r; }
#define OP_LDSB32R(i) { unsigned r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a[i] & 0x1fffcu)); a[i] += r; }
#define OP_LDSB32L(i) { unsigned r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(((a[i] & 0x3ffu) << 7) | ((lane & 31) << 2))); a[i] += r; }
#define OP_LDSB64R(i) { unsigned long long r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a[i] & 0x1fff8u)); a[i] += (unsigned)r; }
#define OP_LDSW64(i) asm volatile("ds_write_b64 %0, %1" : : "v"(((i) * 64 + lane) * 8 + 65536), "v"((unsigned long long)a[i]) : "memory");
#define OP_LDSW64M(i) asm volatile("v_cmp_gt_u32 vcc, 12, %2\n s_and_saveexec_b64 s[30:31], vcc\n ds_write_b64 %0, %1\n s_or_b64 exec, exec, s[30:31]" : : "v"(((i) * 64 + lane) * 8 + 65536), "v"((unsigned long long)a[i]), "v"(lane) : "memory", "vcc", "s30", "s31");
    if (T == 0) { REP64(OP_ADD) }
    if (T == 1) { REP64(OP_MAD24) }
    if (T == 2) { REP64(OP_MULLO) }
    if (T == 3) { REP64(OP_MULHI24) }
    if (T == 4) { REP64(OP_PKMAD) }
    if (T == 5) { REP64(OP_PKMIN) }
    if (T == 6) { REP64(OP_PKSUB) }
    if (T == 7) { REP64(OP_PERM) }
    if (T == 8) { REP64(OP_DOT4) }
    if (T == 9) { REP64(OP_SDWASUB) }
    if (T == 10) { REP64(OP_SDWASHL) }
    if (T == 11) { REP64(OP_BFE) }
    if (T == 12) { REP64(OP_ALIGNBIT) }
    if (T == 13) { REP64(OP_LSHLOR) }
    if (T == 14) { REP64(OP_ADDLSHL) }
    if (T == 15) { REP64(OP_MBCNT) }
    if (T == 16) { REP64(OP_CMP) }
    if (T == 17) { REP64(OP_CMPADDC) }
    if (T == 18) { REP64(OP_ANDOR) }
    if (T == 19) { REP64(OP_BCNT) }
    if (T == 20) { REP64(OP_CNDMASK) }
    if (T == 21) { REP64(OP_SALU) }
    if (T == 22) { REP64(OP_SBCNT) }
    if (T == 23) { REP64(OP_CMPBR) }
    if (T == 24) { REP64(OP_SAVEEXEC) }
    if (T == 25) { REP64(OP_LDSU16) }
    if (T == 26) { REP64(OP_LDSB32R) }
    if (T == 27) { REP64(OP_LDSB32L) }
    if (T == 28) { REP64(OP_LDSB64R) }
    if (T == 29) { REP64(OP_LDSW64) }
    if (T == 30) { REP64(OP_LDSW64M) }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  if (s == 0x12345678u) sink[0] = s + b + c + (unsigned)m;
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static const char *names[] = {"v_add_u32", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32_u24", "v_pk_mad_u16", "v_pk_min_u16", "v_pk_sub_u16",
  "v_perm_b32", "v_dot4_u32_u8", "v_sub_u32_sdwa(byte)", "v_lshlrev_b32_sdwa(word)", "v_bfe_u32", "v_alignbit_b32", "v_lshl_or_b32", "v_add_lshl_u32",
  "v_mbcnt_lo", "v_cmp_ne_u32", "v_cmp+v_addc (2 instr)", "v_and_or_b32", "v_bcnt_u32_b32", "v_cndmask_b32", "s_add_u32", "s_bcnt1+s_add (2 instr)",
  "v_cmp+s_cbranch_vccz+v_add (3 instr)", "v_cmp+saveexec+v_add+s_or (4 instr)", "ds_read_u16 random", "ds_read_b32 random", "ds_read_b32 lane-private bank",
  "ds_read_b64 random", "ds_write_b64 conflict-free", "v_cmp+saveexec+ds_write_b64(12 lanes)+s_or"};

template <int T> void run(int threads, unsigned long long *d_out, unsigned *d_sink, int cus) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_issue<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // 160 KB of LDS per block: exactly one block per CU, as the scan kernels run
  k_issue<T><<<cus, threads, 160 * 1024>>>(d_out, d_sink, 12345u);
  k_issue<T><<<cus, threads, 160 * 1024>>>(d_out, d_sink, 12345u);
  hipDeviceSynchronize();
  int waves = cus * threads / 64;
  std::vector<unsigned long long> h(waves);
  hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  double per_wave = sum / waves / (ITER * 64.0);
  int wps = threads / 256;
  // s_memtime ticks at 100 MHz-derived "shader clock"? report raw ticks; the ratio between rows is what matters
  printf("%-44s waves/SIMD %d: %7.2f ticks per instr-group per wave, %6.2f per SIMD\n", names[T], wps, per_wave, per_wave / wps);
}

template <int T> void run_all(unsigned long long *d_out, unsigned *d_sink, int cus) {
  run<T>(256, d_out, d_sink, cus);
  run<T>(512, d_out, d_sink, cus);
  run<T>(1024, d_out, d_sink, cus);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.name, cus);
  unsigned long long *d_out;
  unsigned *d_sink;
  hipMalloc(&d_out, 8 * 16 * cus);
  hipMalloc(&d_sink, 64);
  run_all<0>(d_out, d_sink, cus); run_all<1>(d_out, d_sink, cus); run_all<2>(d_out, d_sink, cus); run_all<3>(d_out, d_sink, cus);
  run_all<4>(d_out, d_sink, cus); run_all<5>(d_out, d_sink, cus); run_all<6>(d_out, d_sink, cus); run_all<7>(d_out, d_sink, cus);
  run_all<8>(d_out, d_sink, cus); run_all<9>(d_out, d_sink, cus); run_all<10>(d_out, d_sink, cus); run_all<11>(d_out, d_sink, cus);
  run_all<12>(d_out, d_sink, cus); run_all<13>(d_out, d_sink, cus); run_all<14>(d_out, d_sink, cus); run_all<15>(d_out, d_sink, cus);
  run_all<16>(d_out, d_sink, cus); run_all<17>(d_out, d_sink, cus); run_all<18>(d_out, d_sink, cus); run_all<19>(d_out, d_sink, cus);
  run_all<20>(d_out, d_sink, cus); run_all<21>(d_out, d_sink, cus); run_all<22>(d_out, d_sink, cus); run_all<23>(d_out, d_sink, cus);
  run_all<24>(d_out, d_sink, cus); run_all<25>(d_out, d_sink, cus); run_all<26>(d_out, d_sink, cus); run_all<27>(d_out, d_sink, cus);
  run_all<28>(d_out, d_sink, cus); run_all<29>(d_out, d_sink, cus); run_all<30>(d_out, d_sink, cus);
  return 0;
}
