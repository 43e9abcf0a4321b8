// BASELINE config[4], round 3: ViT attention on the BLOCK-SCALED fp8 matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4
// (OCP MX: e4m3 elements, e8m0 scale per 32-element K block; 2.06x the f16 MFMA rate measured on MI355X,
// scripts/mx_probe) -- Q K^T and P V of dino's Attention.forward, reached from the reference at video_attention.py:52.
// Replaces round 2's non-scaled fp8 kernel (vit_attn_fp8.hip: bf16 rate, per-(frame, head) scales from a two-pass
// pre-kernel); the operand images (vit_mx.h) are written by the attn.qkv GEMM epilogue, no pre-pass.
//
// Operand layout of the instruction, measured with exact data (scripts/mx_probe/mx_explore.hip, gpurun_out/mx_explore.txt):
//   lane (r = lane & 31, h = lane >> 5) holds 32 bytes of row r of A (column r of B): bytes 0..15 = k 16 h .. 16 h + 15,
//   bytes 16..31 = k 32 + 16 h .. 32 + 16 h + 15; K block b = k in [32 b, 32 b + 32) takes its e8m0 scale from the scale
//   operand (byte `opsel`) of lane r + 32 b; C/D as every 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
//
// Workgroup = 128 queries of one (frame, head), 4 waves x 32 queries, query on the lane.  Per 64-key tile and wave FOUR matrix
// instructions (the f16 kernel issues 16 of the 32x32x16 form, 2x the matrix cycles):
//   S^T[32 keys x 32 q] = K8 Q8^T, twice: A = the key rows -- MFMA row 8 g + 4 h' + i is fed with key 32 t + 16 h' + 4 g + i of
//       the tile, so that the accumulator registers (g, i) of a lane half h are keys 16 h + 4 g + i: CONSECUTIVE 16 keys;
//   softmax in f32 exactly as vit_attn.hip (running maximum folded into the MFMA's C operand, deferred rescale), P' = 2^7 P <= 2^8
//       rounded to e4m3 (the 2^7 keeps small probabilities above e4m3's subnormal floor and cancels in O / l);
//   O^T[32 d x 32 q] += V8T P'^T, twice (d halves): the 16 converted registers of key block t are bytes 16 t .. 16 t + 15 of the
//       B operand, i.e. K block t = keys 32 t .. 32 t + 31 in natural order, which is what a 32-byte row piece of V^T is: the
//       scale of (d, 32-token block) is constant along the instruction's K block as MX requires.  P's scale is 1.
// Frames are 785 rows apart but V's scale blocks are aligned on the global row index: a frame's keys are walked in tiles that
// start at floor(row0 / 32) * 32 (13 tiles for 785 keys whatever the offset), rows of the neighbouring frames masked to -inf.
#include <type_traits>
#include "mma.h"
#include "vit_mx.h"

#define MXA_D 64
#define MXA_QT 128
#define MXA_KT 64
#define MXA_THR 1.0f       // log2 units: P = exp2(s - m_run) <= 2 between rescales, P' <= 2^8 < 448
#define MXA_PSHIFT 7.0f

typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ f32x16 mfma_mx(const i32x8& a, const i32x8& b, const f32x16& c, int scale_a, int scale_b) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, scale_a, 0, scale_b);
}
// D = A B + C with C and D in DIFFERENT registers (C = the persistent -m_run block): as a builtin hipcc ties D to C for the second
// instruction of a tile and copies the 16 C registers first (8 v_mov_b64 per tile).  Inline asm: the compiler inserts the waitcnt
// for the LDS-loaded operands, but no MFMA hazard padding -- the operands here come from LDS reads or are long-lived, and the
// readers of D sit behind the explicit s_nop block of the softmax.
__device__ __forceinline__ f32x16 mfma_mx_c(const i32x8& a, const i32x8& b, const f32x16& c, int scale_a, int scale_b) {
  f32x16 d;
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]"
               : "=&v"(d) : "v"(a), "v"(b), "v"(c), "v"(scale_a), "v"(scale_b));
  return d;
}
__device__ __forceinline__ i32x8 cat16(uint4 lo, uint4 hi) {
  i32x8 v;
  v[0] = (int)lo.x; v[1] = (int)lo.y; v[2] = (int)lo.z; v[3] = (int)lo.w;
  v[4] = (int)hi.x; v[5] = (int)hi.y; v[6] = (int)hi.z; v[7] = (int)hi.w;
  return v;
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void vit_attn_mx_kernel(
    MxImages m, bf16_t* __restrict__ out, int ntok, int ld_out, int heads, int qblocks, int ngroups) {
  // tiles: 64 rows x 64 B each, 16-B chunk c of row x at chunk position c ^ ((x >> 2) & 3) (conflict-free b128 fragment reads)
  __shared__ __attribute__((aligned(16))) unsigned char Ks[2][MXA_KT * MXA_D];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[2][MXA_D * MXA_KT];
  __shared__ __attribute__((aligned(16))) unsigned char Scl[2][256];         // [buf]: K scales [d block][64 keys], then V scales [key block][64 d]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / qblocks) * 8 + xcd;
  if (group >= ngroups) return;
  const int qb = slot % qblocks;
  const int head = group % heads, frame = group / heads;
  const int q0 = qb * MXA_QT + wv * 32;
  const bool wave_active = q0 < ntok;
  const int64_t row0 = (int64_t)frame * ntok;
  const int64_t start = row0 & ~(int64_t)31;              // first tile row (global): aligned on V's scale blocks
  const int lead = (int)(row0 - start);                   // rows of the previous frame in tile 0
  const int ntiles = (lead + ntok + MXA_KT - 1) / MXA_KT;

  // Q^T (B operand): lane (q = r, h) holds Q8[q][16 h ..] and Q8[q][32 + 16 h ..]; its scale operand the scale of d block h
  i32x8 fq;
  int sq;
  {
    int qr = q0 + r;
    qr = qr < ntok ? qr : ntok - 1;
    const unsigned char* qp = m.q8 + (row0 + qr) * MX_DIM + head * MXA_D + 16 * h;
    fq = cat16(*reinterpret_cast<const uint4*>(qp), *reinterpret_cast<const uint4*>(qp + 32));
    sq = m.sq[(int64_t)(head * 2 + h) * m.rows_alloc + row0 + qr];
  }
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float mrow = 0.f;
  // Row sum l = sum_k P'[k][q] on the MATRIX pipe: one more instruction per tile with an all-ones A operand (every output row is
  // the sum; the loop is bound by vector issue -- PMC: 52 VALU instructions per MFMA, matrix pipe 19 % busy -- and the 32 adds
  // of the row sum were a third of the non-transcendental VALU work).  It sums the e4m3-ROUNDED P', the very weights of O.
  f32x16 lacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) lacc[e] = 0.f;
  i32x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = 0x38383838;     // e4m3 1.0
  f32x16 cinit;                          // -m_run (+ the P' shift) in all 16 registers: the C operand of the Q K^T instructions
#pragma unroll
  for (int e = 0; e < 16; ++e) cinit[e] = MXA_PSHIFT;

  // staging through registers (one 16-byte piece of each tile per thread; wave 0 also four scale bytes per lane), next tile in
  // flight during this one.  One 64-bit workspace base (scalar) + 32-bit per-thread offsets, advanced per tile.
  uint4 kreg, vreg;
  unsigned sreg = 0;
  const int srow = tid >> 2, sc = tid & 3;
  const unsigned char* wsb = m.q8;
  unsigned koffg = (unsigned)((m.k8 - wsb) + (start + srow) * MX_DIM + head * MXA_D + sc * 16);
  unsigned voffg = (unsigned)((m.v8t - wsb) + (int64_t)(head * MXA_D + srow) * m.rows_alloc + start + sc * 16);
  // scales: lanes 0..31 of wave 0 -> K, d block lane >> 4, keys 4 (lane & 15)..; lanes 32..63 -> V, key block (lane >> 4) & 1,
  // d rows 4 (lane & 15)..  (sv is stored [32-token block][384]: both are 4 consecutive bytes)
  unsigned soffg = h == 0 ? (unsigned)((m.sk - wsb) + (int64_t)(head * 2 + (lane >> 4)) * m.rows_alloc + start + 4 * (lane & 15))
                          : (unsigned)((m.sv - wsb) + ((start >> 5) + ((lane >> 4) & 1)) * MX_DIM + head * MXA_D + 4 * (lane & 15));
  const unsigned sstep = h == 0 ? (unsigned)MXA_KT : (unsigned)(2 * MX_DIM);
  auto load_tile = [&]() {
    kreg = *reinterpret_cast<const uint4*>(wsb + koffg);
    vreg = *reinterpret_cast<const uint4*>(wsb + voffg);
    if (wv == 0) sreg = *reinterpret_cast<const unsigned*>(wsb + soffg);
    koffg += MXA_KT * MX_DIM;
    voffg += MXA_KT;
    soffg += sstep;
  };
  const int st_off = srow * 64 + ((sc ^ ((srow >> 2) & 3)) << 4);
  auto store_tile = [&](int buf) {
    *reinterpret_cast<uint4*>(&Ks[buf][st_off]) = kreg;
    *reinterpret_cast<uint4*>(&Vs[buf][st_off]) = vreg;
    if (wv == 0) *reinterpret_cast<unsigned*>(&Scl[buf][4 * lane]) = sreg;
  };
  // fragment addresses.  K: MFMA row r = 8 g + 4 h' + i reads key 16 h' + 4 g + i (+ 32 t); V: row r reads d row r (+ 32 db)
  const int kkey = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
  const int koff0 = kkey * 64 + ((h ^ ((kkey >> 2) & 3)) << 4);          // chunk h of the row; chunk 2 + h sits at ^ 32
  const int voff0 = r * 64 + ((h ^ ((r >> 2) & 3)) << 4);

  auto max3 = [](float a, float b, float c) __attribute__((always_inline)) {
    float d;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
  };
  load_tile();
  store_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the Q fragment too
  __syncthreads();

  auto tile = [&](int kt) __attribute__((always_inline)) {
    const int buf = kt & 1;
    const unsigned char* kt_base = &Ks[buf][0];
    const unsigned char* vt_base = &Vs[buf][0];
    __builtin_amdgcn_s_setprio(1);
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned char* kp = kt_base + t * 32 * 64 + koff0;
      const i32x8 fk = cat16(*reinterpret_cast<const uint4*>(kp), *reinterpret_cast<const uint4*>(kp + ((koff0 & 32) ? -32 : 32)));
      const int sk = Scl[buf][h * 64 + 32 * t + kkey];
      s[t] = mfma_mx_c(fk, fq, cinit, sk, sq);
    }
    __builtin_amdgcn_s_setprio(0);
    // wait states between the matrix write and the VALU reads inside the asm max chain (hipcc does not look into asm)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s[0]), "+v"(s[1]));
    if (kt == 0 || kt == ntiles - 1) {
      // keys of the neighbouring frames (first / last tile): register (g, i) of tile t in lane half h is key 32 t + 16 h + 4 g + i
      const int kv0 = kt * MXA_KT - lead;             // frame-local index of the tile's first key
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kv0 + 32 * t + 16 * h + 4 * (e >> 2) + (e & 3);
          if (key < 0 || key >= ntok) s[t][e] = -1e30f;
        }
    }
    float mx;
    {
      float a[2][5];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 5; ++g) a[t][g] = max3(s[t][3 * g], s[t][3 * g + 1], s[t][3 * g + 2]);
        a[t][0] = max3(a[t][0], a[t][1], s[t][15]);
        a[t][2] = max3(a[t][2], a[t][3], a[t][4]);
      }
      mx = max3(max3(a[0][0], a[0][2], a[1][0]), a[1][2], a[1][2]);
    }
    // s = score - m_run + 7; rescale when the row maximum exceeds 2^(7 + THR)
    const bool first = kt == 0;
    if (__any(first || mx > MXA_PSHIFT + MXA_THR)) {
      float ma, mb;
      lane_swap32(mx, ma, mb);
      mx = fmaxf(ma, mb);
      const float delta = (first || mx > MXA_PSHIFT + MXA_THR) ? mx - MXA_PSHIFT : 0.f;
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      mrow += delta;
#pragma unroll
      for (int e = 0; e < 16; ++e) cinit[e] = MXA_PSHIFT - mrow;
#pragma unroll
      for (int e = 0; e < 16; ++e) lacc[e] *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[t][e] -= delta;
    }
    i32x8 fp;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[t][e] = __builtin_amdgcn_exp2f(s[t][e]);
#pragma unroll
      for (int g = 0; g < 4; ++g) fp[4 * t + g] = (int)mx_cvt4(s[t][4 * g], s[t][4 * g + 1], s[t][4 * g + 2], s[t][4 * g + 3]);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const unsigned char* vp = vt_base + db * 32 * 64 + voff0;
      const i32x8 fv = cat16(*reinterpret_cast<const uint4*>(vp), *reinterpret_cast<const uint4*>(vp + ((voff0 & 32) ? -32 : 32)));
      const int sv = Scl[buf][128 + h * 64 + 32 * db + r];
      o[db] = mfma_mx(fv, fp, o[db], sv, 127);
    }
    lacc = mfma_mx(ones, fp, lacc, 127, 127);
    __builtin_amdgcn_s_setprio(0);
  };
  for (int kt = 0; kt < ntiles - 1; ++kt) {
    load_tile();
    __builtin_amdgcn_sched_barrier(0);
    if (wave_active) tile(kt);
    asm volatile("" : "+v"(kreg.x), "+v"(kreg.y), "+v"(kreg.z), "+v"(kreg.w));
    asm volatile("" : "+v"(vreg.x), "+v"(vreg.y), "+v"(vreg.z), "+v"(vreg.w));
    store_tile((kt & 1) ^ 1);
    __syncthreads();
  }
  if (wave_active) tile(ntiles - 1);
  // ---- normalise and store: lane (q = r, h) holds d = 32 db + 8 g + 4 h + (0..3) in registers 4g..4g+3 of o[db]
  {
    const float inv = 1.f / lacc[0];            // every row of the ones-product holds the full sum over the keys
    const int qr = q0 + r;
    if (qr < ntok) {
      bf16_t* op = out + (row0 + qr) * ld_out + head * MXA_D + 4 * h;
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

// Stand-alone quantiser: the 16-bit qkv tensor [rows][1152] -> the MX images (what the fused attn.qkv GEMM epilogue writes
// directly; this form serves kernel tests and callers that hold a 16-bit qkv).  One workgroup per 32-row block; correctness
// first: 1152 (row-block, column) absmax / convert items per workgroup, V through an LDS transpose.
template <int MODE>
__global__ __launch_bounds__(256) void vit_qkv_mx_kernel(const bf16_t* __restrict__ qkv, MxImages m, int64_t rows, int ld_qkv) {
  __shared__ float vt[32][MX_DIM + 1];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  // q, k: item = (row, plane of 32 columns), 32 rows x 24 planes
  for (int it = tid; it < 32 * 24; it += 256) {
    const int row = it / 24, pl = it % 24, which = pl / 12, p12 = pl % 12;
    const int64_t gr = r0 + row;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      v[c] = gr < rows ? up16<MODE>(qkv[gr * ld_qkv + which * MX_DIM + p12 * 32 + c]) : 0.f;
      amax = fmaxf(amax, fabsf(v[c]));
    }
    float inv;
    const unsigned e = mx_scale_byte(amax, inv);
    unsigned char* dst = (which ? m.k8 : m.q8) + gr * MX_DIM + p12 * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 4) *reinterpret_cast<unsigned*>(dst + c) = mx_cvt4(v[c] * inv, v[c + 1] * inv, v[c + 2] * inv, v[c + 3] * inv);
    (which ? m.sk : m.sq)[(int64_t)p12 * m.rows_alloc + gr] = (unsigned char)e;
  }
  // v: transpose the block through LDS, then item = d row
  for (int it = tid; it < 32 * MX_DIM; it += 256) {
    const int row = it / MX_DIM, c = it % MX_DIM;
    const int64_t gr = r0 + row;
    vt[row][c] = gr < rows ? up16<MODE>(qkv[gr * ld_qkv + 2 * MX_DIM + c]) : 0.f;
  }
  __syncthreads();
  for (int d = tid; d < MX_DIM; d += 256) {
    float amax = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t) amax = fmaxf(amax, fabsf(vt[t][d]));
    float inv;
    const unsigned e = mx_scale_byte(amax, inv);
    unsigned char* dst = m.v8t + (int64_t)d * m.rows_alloc + r0;
#pragma unroll
    for (int t = 0; t < 32; t += 4)
      *reinterpret_cast<unsigned*>(dst + t) = mx_cvt4(vt[t][d] * inv, vt[t + 1][d] * inv, vt[t + 2][d] * inv, vt[t + 3][d] * inv);
    m.sv[(r0 >> 5) * MX_DIM + d] = (unsigned char)e;
  }
}

extern "C" int64_t maavss_vit_attn_mx_ws_bytes(int64_t rows) { return rows > 0 ? mx_ws_bytes(rows) : 0; }

extern "C" int maavss_vit_qkv_mx(const void* qkv, void* ws, int64_t rows, int ld_qkv, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(qkv && ws && rows > 0 && ld_qkv >= 3 * MX_DIM, "vit_qkv_mx: bad arguments");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_qkv_mx: dtype (of qkv) must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(((uintptr_t)ws & 255) == 0, "vit_qkv_mx: ws must be 256-byte aligned");
  MAAVSS_CHECK_ARG(mx_ws_bytes(rows) < (1LL << 32), "vit_qkv_mx: %ld rows: workspace beyond the attention kernel's 32-bit offsets", (long)rows);
  const MxImages m = mx_images(ws, rows);
  const unsigned nb = (unsigned)(m.rows_alloc / 32);        // all allocated row blocks: the tail becomes exact zeros
  if (dtype == MODE_F16) hipLaunchKernelGGL(vit_qkv_mx_kernel<MODE_F16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, m, rows, ld_qkv);
  else hipLaunchKernelGGL(vit_qkv_mx_kernel<MODE_BF16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, m, rows, ld_qkv);
  MAAVSS_LAUNCH_CHECK("vit_qkv_mx_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_attn_mx(const void* ws, void* out, int frames, int ntok, int heads, int ld_out, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(ws && out && frames > 0 && ntok > 0, "vit_attn_mx: bad arguments");
  MAAVSS_CHECK_ARG(heads * MXA_D == MX_DIM && ld_out >= MX_DIM && ld_out % 4 == 0, "vit_attn_mx: built for 6 heads x 64 (ViT-S); bad layout");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_attn_mx: dtype (of out) must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(((uintptr_t)ws & 255) == 0, "vit_attn_mx: ws must be 256-byte aligned");
  const int64_t rows = (int64_t)frames * ntok;
  // the kernel addresses the workspace with 32-bit byte offsets
  MAAVSS_CHECK_ARG(mx_ws_bytes(rows) < (1LL << 32), "vit_attn_mx: %ld rows need a workspace of %ld bytes; the kernel's offsets are 32-bit (launch fewer frames per group)",
                   (long)rows, (long)mx_ws_bytes(rows));
  const MxImages m = mx_images(const_cast<void*>(ws), rows);
  const int qblocks = cdiv(ntok, MXA_QT), ngroups = frames * heads;
  const int nblocks = cdiv(ngroups, 8) * 8 * qblocks;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MODE_F16) hipLaunchKernelGGL(vit_attn_mx_kernel<MODE_F16>, dim3(nblocks), dim3(256), 0, st, m, (bf16_t*)out, ntok, ld_out, heads, qblocks, ngroups);
  else hipLaunchKernelGGL(vit_attn_mx_kernel<MODE_BF16>, dim3(nblocks), dim3(256), 0, st, m, (bf16_t*)out, ntok, ld_out, heads, qblocks, ngroups);
  MAAVSS_LAUNCH_CHECK("vit_attn_mx_kernel");
  return MAAVSS_OK;
}
