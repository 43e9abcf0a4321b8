// K4/K5: multi-head self-attention of the ViT blocks (softmax(Q K^T / 8) V, 6 heads x 64) and the CLS-row
// attention of the last block -- what dino's Attention.forward / get_last_selfattention compute for the
// reference's video_attention.py:52-56.
//
// vit_attn_kernel<MODE>: flash-style, never materialises the N x N scores; MODE = bf16 or IEEE half operands, f32 softmax.
// qkv is the fused projection output [rows][1152] (16-bit) with q pre-scaled by log2(e)/8 in the GEMM epilogue (softmax
// runs on exp2); out is [rows][384] with heads concatenated.  Structure and layout notes sit above the kernel.
// vit_cls_attn_kernel: last block only -- the CLS query against all keys, softmax over N tokens, the CLS
// column dropped (video_attention.py:56): att [frames][6][N-1] f32.
#include <type_traits>
#include "mma.h"

#define ATT_D 64
#define ATT_QT 128
#define ATT_KT 64
#define ATT_THR 5.0f  // log2 units: rescale only when the running maximum grows by more than 2^5
// Ablation switches for measurement builds only (make ablate -> lib/libmaavss_ablate.so; results are WRONG by design):
// bit 0 no exp, 1 no row sum, 2 no 16-bit pack, 3 no max chain, 4 no QK^T MFMAs, 5 no P V MFMAs, 6 no K/V staging after tile 0,
// 7 no LDS fragment reads (fragments read once before the loop)
#ifdef MAAVSS_ATTN_ABLATE
#include <stdlib.h>
#define ABL(bit) ((ABLM >> (bit)) & 1)
#else
#define ABL(bit) 0
#endif

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Workgroup = 128 query rows of one (frame, head), 4 waves x 32 rows, on v_mfma_f32_32x32x16_{bf16,f16}.  (Round 1 ran the
// same algorithm on the 16x16x32 form; both take the same time on MI355X, random data 758 us / zeros 599 us per launch at
// 512 frames x 785 tokens -- the loop is not bound by the MFMA issue hold, DESIGN.md 9 -- this form needs 10 fewer VGPRs and
// halves the MFMA instruction count.)  Per 64-key tile and wave:
//   S^T[64 keys x 32 q] = K Q^T : 2 key blocks x 4 d-steps of 32x32x16, K = A operand (ds_read_b128, chunk ^ ((key>>1)&7):
//                                conflict-free for the b128 lane groups at a 128-B row stride), query on the lane;
//   softmax: 32 scores per lane (the query's other 32 keys sit on lane ^ 32), running maximum folded into the MFMA's C,
//            deferred rescale (one v_permlane32_swap, only when m_run moves);
//   O^T[64 d x 32 q] += V^T P^T : the exponentiated accumulator registers 8s..8s+7 of key block kb ARE the B fragment of
//            k-step 2 kb + s (k slot (h, j) = key 32 kb + 16 s + 8 (j>>2) + 4 h + (j&3)); V^T fragments with the same slot
//            assignment come from two ds_read_b64_tr_b16 (64-B half of a V row ^= (key>>1)&1: every 32-lane half of a
//            transposed read covers all 64 banks once).
// The last key tile runs a single 32-key block when that covers the remaining keys (785 tokens: 17 of 64).
template <int MODE, int ABLM = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void vit_attn_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int ntok, int ld_qkv, int ld_out, int dim, int heads, int qblocks,
    int ngroups) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][ATT_KT * ATT_D];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][ATT_KT * ATT_D];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / qblocks) * 8 + xcd;
  if (group >= ngroups) return;
  const int qb = slot % qblocks;
  const int head = group % heads, frame = group / heads;
  const int q0 = qb * ATT_QT + wv * 32;
  const bool wave_active = q0 < ntok;
  const int64_t row0 = (int64_t)frame * ntok;
  const bf16_t* qbase = qkv + head * ATT_D;
  const bf16_t* kbase = qkv + dim + head * ATT_D;
  const bf16_t* vbase = qkv + 2 * dim + head * ATT_D;

  // Q^T fragments (B operand): lane (q = r, h) holds Q[q][16 ks + 8 h .. +7]
  bf16x8 fq[4];
  {
    int qr = q0 + r;
    qr = qr < ntok ? qr : ntok - 1;
    const bf16_t* qp = qbase + (row0 + qr) * ld_qkv + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fq[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float mrow = 0.f, lrow = 0.f;
  f32x16 cinit;                          // -m_run in all 16 registers: the C operand of every tile's first MFMA
#pragma unroll
  for (int e = 0; e < 16; ++e) cinit[e] = 0.f;

  // K / V staging through registers: the next tile's global loads (wave-uniform frame base + one 32-bit lane offset, clamped
  // to the last row instead of predicated) are issued before the tile's MFMAs and written to LDS after them.  (LDS-DMA
  // staging -- global_load_lds with the swizzles applied on the source side -- was measured 4-8 % SLOWER)
  uint4 kreg[2], vreg[2];
  const char* kframe = reinterpret_cast<const char*>(kbase + row0 * ld_qkv);
  const char* vframe = reinterpret_cast<const char*>(vbase + row0 * ld_qkv);
  unsigned ldoff[2], ldmax;
#pragma unroll
  for (int it = 0; it < 2; ++it) ldoff[it] = (unsigned)((it * 32 + (tid >> 3)) * ld_qkv + (tid & 7) * 8) * 2u;
  ldmax = (unsigned)((ntok - 1) * ld_qkv + (tid & 7) * 8) * 2u;
  const unsigned tile_step = (unsigned)(ATT_KT * ld_qkv) * 2u;
  auto load_tile = [&]() {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const unsigned off = ldoff[it] < ldmax ? ldoff[it] : ldmax;
      kreg[it] = *reinterpret_cast<const uint4*>(kframe + off);
      vreg[it] = *reinterpret_cast<const uint4*>(vframe + off);
      ldoff[it] += tile_step;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = it * 256 + tid, key = idx >> 3, c = idx & 7;
      *reinterpret_cast<uint4*>(&Ks[buf][key * ATT_D + ((c ^ ((key >> 1) & 7)) * 8)]) = kreg[it];
      *reinterpret_cast<uint4*>(&Vs[buf][key * ATT_D + (((c >> 2) ^ ((key >> 1) & 1)) * 32) + (c & 3) * 8]) = vreg[it];
    }
  };
  // per-lane LDS element offsets; key block / k-step / buffer displacements are compile-time or wave-uniform
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * ATT_D + (((2 * ks + h) ^ ((r >> 1) & 7)) * 8);
  const int i16 = lane & 15, g2 = (lane >> 4) & 1, xb = (i16 >> 3) & 1;
  const int vrow = (4 * h + (i16 >> 2)) * ATT_D + 16 * g2 + 4 * (i16 & 3);
  const int voff[2] = {vrow + xb * 32, vrow + (1 ^ xb) * 32};

  const int ntiles = (ntok + ATT_KT - 1) / ATT_KT;
  const bool last_half = ntok - (ntiles - 1) * ATT_KT <= 32;   // the last tile's keys fit one 32-key block
  load_tile();
  store_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the Q fragments too, so that no wait on them lands inside the loop
  __syncthreads();
  // volatile: keeps the chain behind the wait-state asm above (volatile asm statements are not reordered among themselves)
  auto max3 = [](float a, float b, float c) __attribute__((always_inline)) {
    float d;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
  };
  auto tile = [&](int kt, auto last_c, auto nkb_c) __attribute__((always_inline)) {
    constexpr int NKB = decltype(nkb_c)::value;
    const int buf = kt & 1, kv0 = kt * ATT_KT;
    const bf16_t* kt_base = &Ks[buf][0];
    const bf16_t* vt_base = &Vs[buf][0];
    __builtin_amdgcn_s_setprio(1);
    // The running maximum enters as the first MFMA's C operand (s' = score - m_run, no subtraction pass) from a register
    // block that only changes when m_run does: C and D of an MFMA may be different registers, so the 32 v_mov per tile that a
    // re-initialised accumulator costs (15 % of the loop's vector issue) are not spent.
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fk;
        if (ABL(7)) fk = fq[ks]; else fk = *reinterpret_cast<const bf16x8*>(kt_base + kb * 32 * ATT_D + koff[ks]);
        if (ABL(4)) { if (ks == 0) s[kb] = cinit; asm volatile("" : "+v"(s[kb]), "+v"(fk)); }
        else if (ks == 0) s[kb] = Mma32<MODE>::mma3(fk, fq[0], cinit);
        else Mma32<MODE>::mma(s[kb], fk, fq[ks]);
      }
    }
    if (decltype(last_c)::value && kv0 + NKB * 32 > ntok) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= ntok) s[kb][e] = -1e30f;
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- online softmax (base 2), query on the lane; this lane's NKB * 16 keys
    // hipcc's hazard recogniser does not look inside asm statements: the v_max3 chain below reads MFMA results, so the
    // wait states between an XDL write and a VALU read of the same registers (up to 18 for the 16-pass form) are spent here
    // explicitly -- without them the last k-step's contribution is missing from some maxima (no fault: a missed rescale,
    // i.e. P up to 2^31 -- invisible in bf16, infinite in IEEE half).
    if constexpr (NKB == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s[0]), "+v"(s[1]));
    else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s[0]));
    float mx;
    if (ABL(3)) mx = s[0][0];
    else {
      float a[NKB][5];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int g = 0; g < 5; ++g) a[kb][g] = max3(s[kb][3 * g], s[kb][3 * g + 1], s[kb][3 * g + 2]);
        a[kb][0] = max3(a[kb][0], a[kb][1], s[kb][15]);
        a[kb][2] = max3(a[kb][2], a[kb][3], a[kb][4]);
      }
      if constexpr (NKB == 2) mx = max3(max3(a[0][0], a[0][2], a[1][0]), a[1][2], a[1][2]);
      else mx = max3(a[0][0], a[0][2], a[0][2]);
    }
    const bool first = kt == 0;
    if (__any(first || mx > ATT_THR)) {
      float ma, mb;
      lane_swap32(mx, ma, mb);
      mx = fmaxf(ma, mb);
      const float delta = (first || mx > ATT_THR) ? mx : 0.f;
      const float alpha = fast_exp2(-delta);
      mrow += delta;
#pragma unroll
      for (int e = 0; e < 16; ++e) cinit[e] = -mrow;
      lrow *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[kb][e] -= delta;
    }
    bf16x8 fp[NKB][2];
    {
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = ABL(0) ? s[kb][e] : fast_exp2(s[kb][e]);
          s[kb][e] = p;
          if (!ABL(1) || e == 0) sum += p;
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const uint4 u = make_uint4(pack2<MODE>(s[kb][8 * sx], s[kb][8 * sx + 1]), pack2<MODE>(s[kb][8 * sx + 2], s[kb][8 * sx + 3]),
                                     pack2<MODE>(s[kb][8 * sx + 4], s[kb][8 * sx + 5]), pack2<MODE>(s[kb][8 * sx + 6], s[kb][8 * sx + 7]));
          if (ABL(2)) { const uint4 u2 = make_uint4(__float_as_uint(s[kb][8 * sx]), __float_as_uint(s[kb][8 * sx + 2]), __float_as_uint(s[kb][8 * sx + 4]), __float_as_uint(s[kb][8 * sx + 6])); fp[kb][sx] = __builtin_bit_cast(bf16x8, u2); }
          else fp[kb][sx] = __builtin_bit_cast(bf16x8, u);
        }
      }
      lrow += sum;
    }
    __builtin_amdgcn_s_setprio(1);
    // ---- O^T += V^T P^T
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16_t* a0 = vt_base + (kb * 32 + sx * 16) * ATT_D + voff[db];
          bf16x8 fv;
          if (ABL(7)) fv = fq[sx + 2 * db];
          else {
            const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0));
            const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0 + 8 * ATT_D));
            fv = concat4(h0, h1);
          }
          if (ABL(5)) asm volatile("" : "+v"(o[db]) : "v"(fv), "v"(fp[kb][sx]));
          else Mma32<MODE>::mma(o[db], fv, fp[kb][sx]);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  for (int kt = 0; kt < ntiles - 1; ++kt) {
    if (!ABL(6)) load_tile();
    __builtin_amdgcn_sched_barrier(0);   // keep the loads up here: their latency is covered by the tile's MFMAs
    if (wave_active) tile(kt, std::false_type{}, std::integral_constant<int, 2>{});
    // the LDS writes stay below the tile's MFMAs (hipcc otherwise merges them into the load block above)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      asm volatile("" : "+v"(kreg[it].x), "+v"(kreg[it].y), "+v"(kreg[it].z), "+v"(kreg[it].w));
      asm volatile("" : "+v"(vreg[it].x), "+v"(vreg[it].y), "+v"(vreg[it].z), "+v"(vreg[it].w));
    }
    if (!ABL(6)) store_tile((kt & 1) ^ 1);
    __syncthreads();
  }
  if (wave_active) {
    if (last_half) tile(ntiles - 1, std::true_type{}, std::integral_constant<int, 1>{});
    else tile(ntiles - 1, std::true_type{}, std::integral_constant<int, 2>{});
  }
  // ---- normalise and store: lane (q = r, h) holds d = 32 db + 8 g + 4 h + (0..3) in registers 4g..4g+3 of o[db]
  {
    float la, lb;
    lane_swap32(lrow, la, lb);
    const float inv = 1.f / (la + lb);
    const int qr = q0 + r;
    if (qr < ntok) {
      bf16_t* op = out + (row0 + qr) * ld_out + head * ATT_D + 4 * h;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 u;
          u.x = pack2<MODE>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          u.y = pack2<MODE>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + 32 * db + 8 * g) = u;
        }
    }
  }
}

// one block per (frame, head): scores of the CLS query against all tokens (q pre-scaled by log2(e)/8 -> exp2)
template <int MODE>
__global__ __launch_bounds__(256) void vit_cls_attn_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ att, int ntok,
                                                           int ld_qkv, int dim) {
  extern __shared__ float sc[];  // [ntok]
  __shared__ float qv[ATT_D];
  __shared__ float red[4];
  const int tid = threadIdx.x, head = blockIdx.x, frame = blockIdx.y;
  const int64_t row0 = (int64_t)frame * ntok;
  if (tid < ATT_D) qv[tid] = up16<MODE>(qkv[row0 * ld_qkv + head * ATT_D + tid]);
  __syncthreads();
  float mx = -1e30f;
  for (int j = tid; j < ntok; j += 256) {
    const bf16_t* kp = qkv + (row0 + j) * ld_qkv + dim + head * ATT_D;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 u = *reinterpret_cast<const uint4*>(kp + c * 8);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = fmaf(up16<MODE>((unsigned short)(w[e] & 0xffffu)), qv[c * 8 + 2 * e], acc);
        acc = fmaf(up16<MODE>((unsigned short)(w[e] >> 16)), qv[c * 8 + 2 * e + 1], acc);
      }
    }
    sc[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < ntok; j += 256) {
    const float p = exp2f(sc[j] - mx);
    sc[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  float* ap = att + ((int64_t)frame * gridDim.x + head) * (ntok - 1);
  for (int j = tid + 1; j < ntok; j += 256) ap[j - 1] = sc[j] * inv;
}

extern "C" int maavss_vit_attn(const void* qkv, void* out, int frames, int ntok, int heads, int ld_qkv, int ld_out, int dtype,
                               void* stream) {
  MAAVSS_CHECK_ARG(qkv && out && frames > 0 && ntok > 0, "vit_attn: bad arguments");
  MAAVSS_CHECK_ARG(heads >= 1 && ld_qkv >= 3 * heads * ATT_D && ld_out >= heads * ATT_D && ld_qkv % 8 == 0 && ld_out % 4 == 0,
                   "vit_attn: bad layout");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_attn: dtype must be 0 (bf16) or 2 (f16)");
  const int qblocks = cdiv(ntok, ATT_QT), ngroups = frames * heads;
  const int nblocks = cdiv(ngroups, 8) * 8 * qblocks;
  const bf16_t* q = (const bf16_t*)qkv;
  bf16_t* o = (bf16_t*)out;
  hipStream_t st = (hipStream_t)stream;
#ifdef MAAVSS_ATTN_ABLATE
  {
    const char* e = getenv("MAAVSS_ATTN_ABL");
    const int m = e ? atoi(e) : 0;
#define ABL_CASE(M) case M: hipLaunchKernelGGL((vit_attn_kernel<MODE_F16, M>), dim3(nblocks), dim3(256), 0, st, q, o, ntok, ld_qkv, ld_out, heads * ATT_D, heads, qblocks, ngroups); break;
    switch (m) { ABL_CASE(0) ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(7) ABL_CASE(8) ABL_CASE(15) ABL_CASE(16) ABL_CASE(32) ABL_CASE(48) ABL_CASE(64) ABL_CASE(128) ABL_CASE(192) ABL_CASE(207) ABL_CASE(240)
      default: MAAVSS_CHECK_ARG(false, "vit_attn (ablation build): mask %d not instantiated", m); }
#undef ABL_CASE
    MAAVSS_LAUNCH_CHECK("vit_attn_kernel");
    return MAAVSS_OK;
  }
#endif
  if (dtype == MODE_F16)
    hipLaunchKernelGGL(vit_attn_kernel<MODE_F16>, dim3(nblocks), dim3(256), 0, st, q, o, ntok, ld_qkv, ld_out, heads * ATT_D, heads, qblocks, ngroups);
  else
    hipLaunchKernelGGL(vit_attn_kernel<MODE_BF16>, dim3(nblocks), dim3(256), 0, st, q, o, ntok, ld_qkv, ld_out, heads * ATT_D, heads, qblocks, ngroups);
  MAAVSS_LAUNCH_CHECK("vit_attn_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_cls_attn(const void* qkv, float* att, int frames, int ntok, int heads, int ld_qkv, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(qkv && att && frames > 0 && ntok > 1, "vit_cls_attn: bad arguments");
  MAAVSS_CHECK_ARG((size_t)ntok * 4 <= 60 * 1024, "vit_cls_attn: too many tokens for the LDS score buffer");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_cls_attn: dtype must be 0 (bf16) or 2 (f16)");
  if (dtype == MODE_F16)
    hipLaunchKernelGGL(vit_cls_attn_kernel<MODE_F16>, dim3(heads, frames), dim3(256), ntok * sizeof(float), (hipStream_t)stream,
                       (const bf16_t*)qkv, att, ntok, ld_qkv, heads * ATT_D);
  else
    hipLaunchKernelGGL(vit_cls_attn_kernel<MODE_BF16>, dim3(heads, frames), dim3(256), ntok * sizeof(float), (hipStream_t)stream,
                       (const bf16_t*)qkv, att, ntok, ld_qkv, heads * ATT_D);
  MAAVSS_LAUNCH_CHECK("vit_cls_attn_kernel");
  return MAAVSS_OK;
}
