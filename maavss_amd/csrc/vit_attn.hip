// K4/K5: multi-head self-attention of the ViT blocks (softmax(Q K^T / 8) V, 6 heads x 64) and the CLS-row
// attention of the last block -- what dino's Attention.forward / get_last_selfattention compute for the
// reference's video_attention.py:52-56.
//
// vit_attn_kernel: flash-style, never materialises the N x N scores.  Workgroup = 128 query rows of one
// (frame, head); each of the 4 waves owns 32 rows (two 16-row tiles).  Per 64-key tile:
//   S^T = K Q^T  on v_mfma_f32_16x16x32_bf16 with K as the A operand: the accumulator then has the QUERY on the
//         lane and 16 keys in registers, so the softmax row maximum is a register reduction plus two cross-lane
//         exchanges (lanes l, l^16, l^32, l^48), and
//   O^T += V^T P^T takes the exponentiated accumulator, packed to bf16, DIRECTLY as its B operand (the MFMA k
//         slot (g, e) is assigned to key 32*ks + 16*(e>>2) + 4*g + (e&3), and V^T fragments are read with the same
//         assignment by ds_read_b64_tr_b16) -- P never goes through LDS.
// K/V tiles are double-buffered in LDS (K XOR-swizzled per 16-B chunk for ds_read_b128, V per 32-B granule for
// the transposed reads: both conflict-free for the hardware's lane groups); the next tile's global loads (wave-uniform
// frame base + one 32-bit lane offset, clamped to the last row instead of predicated) are issued before the MFMAs and
// written after them; the last key tile, the only one with padded keys, is peeled off the loop.  The row maximum is
// taken per LANE (v_max3 chain); the exchange across a query's four lanes only runs when m_run has to move.  The two
// MFMA phases run at s_setprio 1 so that the other resident waves' softmax VALU fills their issue gaps.
// qkv is the fused projection output [rows][1152] bf16 with q pre-scaled by log2(e)/8 in the GEMM epilogue
// (softmax runs on exp2); out is [rows][384] bf16 with heads concatenated.
// vit_cls_attn_kernel: last block only -- the CLS query against all keys, softmax over N tokens, the CLS
// column dropped (video_attention.py:56): att [frames][6][N-1] f32.
#include <type_traits>
#include "mma.h"

#define ATT_D 64
#define ATT_QT 128
#define ATT_KT 64
#define ATT_THR 5.0f  // log2 units: rescale only when the running maximum grows by more than 2^5

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void vit_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int ntok,
                                                       int ld_qkv, int ld_out, int dim, int heads, int qblocks,
                                                       int ngroups) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][ATT_KT * ATT_D];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][ATT_KT * ATT_D];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, gq = lane >> 4;
  // XCD-aware mapping: the QB query blocks of one (frame, head) share that pair's K/V (200 KB); blocks are dealt
  // round-robin over the 8 XCDs, so give the QB consecutive slots of ONE XCD to the same (frame, head) -- K/V are
  // then fetched from HBM once and re-read from that XCD's L2 (measured 4.4 GB -> see profiles/ per launch).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / qblocks) * 8 + xcd;  // (frame, head) pair index
  if (group >= ngroups) return;                  // padding blocks of the last round
  const int qb = slot % qblocks;
  const int head = group % heads, frame = group / heads;
  const int q0 = qb * ATT_QT + wv * 32;
  const bool wave_active = q0 < ntok;
  const int64_t row0 = (int64_t)frame * ntok;
  const bf16_t* qbase = qkv + head * ATT_D;
  const bf16_t* kbase = qkv + dim + head * ATT_D;
  const bf16_t* vbase = qkv + 2 * dim + head * ATT_D;

  // Q^T fragments (B operand): lane (q = l16, gq) holds Q[q][32 ks + 8 gq .. +7]
  bf16x8 fq[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int qr = q0 + t * 16 + l16;
    qr = qr < ntok ? qr : ntok - 1;  // rows past the end are clamped; their results are never stored
    const bf16_t* qp = qbase + (row0 + qr) * ld_qkv;
    fq[t][0] = *reinterpret_cast<const bf16x8*>(qp + gq * 8);
    fq[t][1] = *reinterpret_cast<const bf16x8*>(qp + 32 + gq * 8);
  }
  f32x4 o[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int d = 0; d < 4; ++d) o[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrow[2] = {0.f, 0.f}, lrow[2] = {0.f, 0.f};   // the first tile rebases m_run unconditionally

  // staging map: thread -> (key = idx>>3, 16-B chunk c = idx&7) for idx = tid and tid + 256
  uint4 kreg[2], vreg[2];
  // K / V addresses: a wave-uniform frame base (SGPRs) plus a 32-bit byte offset per thread that advances by one key
  // tile per iteration (one frame's qkv is ntok * ld_qkv * 2 bytes, far below 4 GB)
  const char* kframe = reinterpret_cast<const char*>(kbase + row0 * ld_qkv);
  const char* vframe = reinterpret_cast<const char*>(vbase + row0 * ld_qkv);
  // Rows past the sequence end are read from the last real row instead (finite values; their scores are masked and
  // their probabilities are exactly 0), so the loads need no predicate and the registers no zero fill.
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
      *reinterpret_cast<uint4*>(&Ks[buf][key * ATT_D + ((c ^ (key & 7)) * 8)]) = kreg[it];
      *reinterpret_cast<uint4*>(&Vs[buf][key * ATT_D + (((c >> 1) ^ ((key >> 1) & 3)) * 16) + (c & 1) * 8]) = vreg[it];
    }
  };

  const int ntiles = (ntok + ATT_KT - 1) / ATT_KT;
  load_tile();
  store_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the Q fragments too, so that no wait on them lands inside the loop
  __syncthreads();
  // one key tile of this wave's 32 queries; `last_c` = the final tile (the only one that can hold padded keys)
  auto tile = [&](int kt, auto last_c) __attribute__((always_inline)) {
    const int buf = kt & 1, kv0 = kt * ATT_KT;
    __builtin_amdgcn_s_setprio(1);
    // ---- S^T = K Q^T : s[t][nt][r] = score(key 16 nt + 4 gq + r, query t*16 + l16)
    // The running row maximum goes in as the MFMA's C operand (s' = score - m_run), so no subtraction pass.
    f32x4 s[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[0][nt] = f32x4{-mrow[0], -mrow[0], -mrow[0], -mrow[0]};
      s[1][nt] = f32x4{-mrow[1], -mrow[1], -mrow[1], -mrow[1]};
      const int key = nt * 16 + l16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 fk = *reinterpret_cast<const bf16x8*>(&Ks[buf][key * ATT_D + (((ks * 4 + gq) ^ (key & 7)) * 8)]);
        Mma<MODE_BF16>::mma(s[0][nt], fk, fq[0][ks]);
        Mma<MODE_BF16>::mma(s[1][nt], fk, fq[1][ks]);
      }
    }
    if (decltype(last_c)::value && kv0 + ATT_KT > ntok) {  // last, partial tile: mask the padded keys
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kv0 + nt * 16 + gq * 4 + r >= ntok) { s[0][nt][r] = -1e30f; s[1][nt][r] = -1e30f; }
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- online softmax (base 2), query on the lane
    // Deferred rescale: m_run only moves when the tile maximum exceeds it by more than ATT_THR (then P <= 2^ATT_THR,
    // harmless in f32 / bf16); the O / l rescale is a rare wave-uniform branch instead of 32 multiplies per tile.
    bf16x8 fp[2][2];
    float mx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // 16 scores per lane and query: eight v_max3_f32 (three inputs per instruction; written as asm so that hipcc does
      // not put a canonicalising v_max in front of every MFMA output)
      auto max3 = [](float a, float b, float c) __attribute__((always_inline)) {
        float d;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
      };
      const float a0 = max3(s[t][0][0], s[t][0][1], s[t][0][2]), a1 = max3(s[t][0][3], s[t][1][0], s[t][1][1]);
      const float a2 = max3(s[t][1][2], s[t][1][3], s[t][2][0]), a3 = max3(s[t][2][1], s[t][2][2], s[t][2][3]);
      const float a4 = max3(s[t][3][0], s[t][3][1], s[t][3][2]);
      mx[t] = max3(max3(a0, a1, a2), a3, max3(a4, s[t][3][3], s[t][3][3]));   // this lane's 16 keys only
    }
    const bool first = kt == 0;
    // The maximum over a query's four lanes is only needed when m_run moves: some lane exceeding the threshold is the
    // same condition as some query exceeding it, so the steady state pays no cross-lane exchange at all.
    if (__any(first || mx[0] > ATT_THR || mx[1] > ATT_THR)) {
      mx[0] = rows4_max(mx[0]);
      mx[1] = rows4_max(mx[1]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float delta = (first || mx[t] > ATT_THR) ? mx[t] : 0.f;   // first tile: rebase in either direction
        const float alpha = fast_exp2(-delta);
        mrow[t] += delta;
        lrow[t] *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[t][d][r] *= alpha;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][nt][r] -= delta;
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(s[t][nt][r]);
          s[t][nt][r] = p;
          sum += p;
        }
      lrow[t] += sum;  // per-lane partial; the 4 lanes of a query are summed at the end
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned w0 = pack_bf2(s[t][2 * ks][0], s[t][2 * ks][1]), w1 = pack_bf2(s[t][2 * ks][2], s[t][2 * ks][3]);
        const unsigned w2 = pack_bf2(s[t][2 * ks + 1][0], s[t][2 * ks + 1][1]), w3 = pack_bf2(s[t][2 * ks + 1][2], s[t][2 * ks + 1][3]);
        const uint4 u = make_uint4(w0, w1, w2, w3);
        fp[t][ks] = __builtin_bit_cast(bf16x8, u);
      }
    }
    __builtin_amdgcn_s_setprio(1);
    // ---- O^T += V^T P^T : o[t][dt][r] = O(query t*16 + l16, d = 16 dt + 4 gq + r)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 h[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int key = ks * 32 + hh * 16 + gq * 4 + (l16 >> 2);
          const bf16_t* a = &Vs[buf][key * ATT_D + ((dt ^ ((key >> 1) & 3)) * 16) + (l16 & 3) * 4];
          h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
        }
        const bf16x8 fv = concat4(h[0], h[1]);
        Mma<MODE_BF16>::mma(o[0][dt], fv, fp[0][ks]);
        Mma<MODE_BF16>::mma(o[1][dt], fv, fp[1][ks]);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // A wave whose 32 query rows all lie past the sequence end (ntok = 785: three of the four waves of the last query
  // block, 11 % of all waves) only helps staging K / V: its MFMA and softmax issue slots go to the other workgroups
  // resident on the SIMD.
  for (int kt = 0; kt < ntiles - 1; ++kt) {
    load_tile();
    __builtin_amdgcn_sched_barrier(0);   // keep the loads up here: their latency is covered by the tile's MFMAs
    if (wave_active) tile(kt, std::false_type{});
    // the LDS writes stay below the tile's MFMAs (hipcc otherwise merges them into the load block above)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      asm volatile("" : "+v"(kreg[it].x), "+v"(kreg[it].y), "+v"(kreg[it].z), "+v"(kreg[it].w));
      asm volatile("" : "+v"(vreg[it].x), "+v"(vreg[it].y), "+v"(vreg[it].z), "+v"(vreg[it].w));
    }
    store_tile((kt & 1) ^ 1);
    __syncthreads();
  }
  if (wave_active) tile(ntiles - 1, std::true_type{});
  // ---- normalise and store: lane holds d = 16 dt + 4 gq + (0..3) of its query
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float l = lrow[t];
    l = rows4_sum(l);
    const float inv = 1.f / l;
    const int qr = q0 + t * 16 + l16;
    if (qr < ntok) {
      bf16_t* op = out + (row0 + qr) * ld_out + head * ATT_D + gq * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 u;
        u.x = pack_bf2(o[t][dt][0] * inv, o[t][dt][1] * inv);
        u.y = pack_bf2(o[t][dt][2] * inv, o[t][dt][3] * inv);
        *reinterpret_cast<uint2*>(op + dt * 16) = u;
      }
    }
  }
}

// one block per (frame, head): scores of the CLS query against all tokens (q pre-scaled by log2(e)/8 -> exp2)
__global__ __launch_bounds__(256) void vit_cls_attn_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ att, int ntok,
                                                           int ld_qkv, int dim) {
  extern __shared__ float sc[];  // [ntok]
  __shared__ float qv[ATT_D];
  __shared__ float red[4];
  const int tid = threadIdx.x, head = blockIdx.x, frame = blockIdx.y;
  const int64_t row0 = (int64_t)frame * ntok;
  if (tid < ATT_D) qv[tid] = bf2f(qkv[row0 * ld_qkv + head * ATT_D + tid]);
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
        acc = fmaf(__uint_as_float(w[e] << 16), qv[c * 8 + 2 * e], acc);
        acc = fmaf(__uint_as_float(w[e] & 0xffff0000u), qv[c * 8 + 2 * e + 1], acc);
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

extern "C" int maavss_vit_attn(const void* qkv, void* out, int frames, int ntok, int heads, int ld_qkv, int ld_out,
                               void* stream) {
  MAAVSS_CHECK_ARG(qkv && out && frames > 0 && ntok > 0, "vit_attn: bad arguments");
  MAAVSS_CHECK_ARG(heads >= 1 && ld_qkv >= 3 * heads * ATT_D && ld_out >= heads * ATT_D && ld_qkv % 8 == 0 && ld_out % 4 == 0,
                   "vit_attn: bad layout");
  const int qblocks = cdiv(ntok, ATT_QT), ngroups = frames * heads;
  const int nblocks = cdiv(ngroups, 8) * 8 * qblocks;
  hipLaunchKernelGGL(vit_attn_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, ntok,
                     ld_qkv, ld_out, heads * ATT_D, heads, qblocks, ngroups);
  MAAVSS_LAUNCH_CHECK("vit_attn_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_cls_attn(const void* qkv, float* att, int frames, int ntok, int heads, int ld_qkv, void* stream) {
  MAAVSS_CHECK_ARG(qkv && att && frames > 0 && ntok > 1, "vit_cls_attn: bad arguments");
  MAAVSS_CHECK_ARG((size_t)ntok * 4 <= 60 * 1024, "vit_cls_attn: too many tokens for the LDS score buffer");
  hipLaunchKernelGGL(vit_cls_attn_kernel, dim3(heads, frames), dim3(256), ntok * sizeof(float), (hipStream_t)stream,
                     (const bf16_t*)qkv, att, ntok, ld_qkv, heads * ATT_D);
  MAAVSS_LAUNCH_CHECK("vit_cls_attn_kernel");
  return MAAVSS_OK;
}
