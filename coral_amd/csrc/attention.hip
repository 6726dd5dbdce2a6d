// Fused (flash-style) multi-head attention for gfx950: forward and backward without ever writing the
// [T, T] score / probability matrices to HBM.  Replaces SDPA at
// $TF/models/wav2vec2/modeling_wav2vec2.py:438-463,529-543 and $TF/models/whisper/modeling_whisper.py:215-238.
//
// Shapes: Q [B, Tq, *] / K, V [B, Tk, *] bf16 with row strides ldq / ldk / ldv and head h at column
// offset h*hd (so q, k, v can live in one fused [B*T, 3d] projection output); head_dim hd is any
// multiple of 8 up to 128 (64, 80 and 120 on this path).
//
// Structure (all three kernels): one 256-thread workgroup = 4 waves x 16 rows (queries or keys); the
// rows of the *other* side stream through LDS in tiles staged by global_load_lds (two images per tile
// where needed: a K-major one read with ds_read_b128 for the Q.K^T / dO.V^T products and an MN-major one
// read with ds_read_b64_tr_b16 for the products that contract over the tile's rows).  Score tiles are
// computed in the orientation whose accumulator registers ARE the next MFMA's A operand (the rows of a
// 32-row step are permuted so that a lane group ends up with 8 consecutive contraction indices), so
// probabilities never leave registers.  The forward is one pass with a running maximum (online softmax);
// the backward kernels recompute probabilities from the saved log-sum-exp.
#include "common.h"
#include <type_traits>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __attribute__((aligned(16))) uint32_t g_attn_zero_page[4];

__device__ __forceinline__ void glds16a(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ---- LDS images -----------------------------------------------------------------------------------
// K-major image: [HDPV/64 chunks][ROWS][64 dims], 128-B rows, 16-B chunk index XOR ((row>>1)&7).
// NW = number of waves that share the load (4 = the whole workgroup, 1 = a wave-private image)
// Chunk swizzle of the K-major images: the 16-byte chunk c of row r sits at position c ^ kswz(r) of its 128-byte row.
// Two read patterns share an image: operand rows (ds_read_b128 of row 32 s + 8 (r >> 2) + 4 bb + (r & 3), chunk 4 ks + g:
// the instruction's four lane groups are NOT contiguous - {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS) - so one
// group holds 16 rows of two lane groups g) and transposed reads (ds_read_b64_tr_b16 of rows 8 g + q, two 32-lane
// groups, each lane pair a chunk pair).  kswz = (r0, r0 ^ r1, r3) is one of the XOR-linear maps of the row bits for
// which a bank simulation of every read of a step, with the hardware's lane groups, finds no conflict at all
// (tools/archive/dev_lds_swizzle.py); the round-1 map (r >> 1) & 7 cost an extra LDS cycle on every read of either kind
// (SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE in dQ and dK|dV).  LDS-DMA writes are linear and do not care.
// MN-major images ([rows][HDPV dims], read only transposed): XOR value of the chunk index of row kr.  A 32-lane group of
// ds_read_b64_tr_b16 covers rows 8 g + q (q = 0..3, g in {0, 1}) of a 32-row block, each lane pair a chunk PAIR: the
// four even rows need four distinct values of bits 1-2 (and bit 3 where a row has 16 chunks).  With 8 chunks per row
// (head_dim <= 64) the round-1 form (2 q | 8 (g & 1)) & 7 lost the g bit: two-way conflicts on every value-fragment read
// of the forward kernels (SQ_LDS_BANK_CONFLICT 37 % of SQ_LDS_IDX_ACTIVE).
template <int PC>
__device__ __forceinline__ int mnswz(int kr) {
#ifndef CA_KSWZ_OLD
  if (PC == 8) return (((kr >> 1) & 1) << 1) | (((kr >> 3) & 1) << 2);
#endif
  return ((((kr & 3) | (((kr >> 3) & 1) << 2))) << 1) & (PC - 1);
}
#ifndef CA_KSWZ_OLD
__device__ __forceinline__ int kswz(int r) { return (r & 1) | (((r ^ (r >> 1)) & 1) << 1) | (((r >> 3) & 1) << 2); }
#else
__device__ __forceinline__ int kswz(int r) { return (r >> 1) & 7; }
#endif
template <int ROWS, int HDPV, int NW = 4>
__device__ __forceinline__ void load_kmajor_image(char* lds, const unsigned short* base, int64_t ld,
                                                  int row0, int nrows, int hd, int wave, int lane) {
  constexpr int NCH = HDPV / 64;
  constexpr int IPW = ROWS / 8 / NW;  // instructions per wave per chunk (ROWS*128 B / 1 KiB / NW waves)
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int inst = wave * IPW + i;
      const int r = inst * 8 + (lane >> 3);
      const int cc = (lane & 7) ^ kswz(r);
      const int dim = c * 64 + cc * 8;
      int row = row0 + r;
      row = row < nrows ? row : nrows - 1;
      const void* src = dim < hd ? (const void*)(base + (int64_t)row * ld + dim) : (const void*)g_attn_zero_page;
      glds16a(src, lds + c * ROWS * 128 + inst * 1024);
    }
}
// MN-major image: [ROWS][HDPV dims], chunk index XOR ((swz(row) << 1) & (PC-1)); rows >= nrows are zero.
template <int ROWS, int HDPV, int NW = 4>
__device__ __forceinline__ void load_mnmajor_image(char* lds, const unsigned short* base, int64_t ld,
                                                   int row0, int nrows, int hd, int wave, int lane) {
  constexpr int PC = HDPV / 8;
  constexpr int RPI = 64 / PC;
  constexpr int IPW = ROWS * HDPV * 2 / 1024 / NW;
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int inst = wave * IPW + i;
    const int kr = inst * RPI + lane / PC;
    const int c = (lane % PC) ^ mnswz<PC>(kr);
    const int dim = c * 8;
    const int row = row0 + kr;
    const void* src = (row < nrows && dim < hd) ? (const void*)(base + (int64_t)row * ld + dim)
                                                : (const void*)g_attn_zero_page;
    glds16a(src, lds + inst * 1024);
  }
}
template <int ROWS>
__device__ __forceinline__ bf16x8_t kimg_frag(const char* img, int row, int ks, int lane) {
  const int cc = (4 * (ks & 1) + (lane >> 4)) ^ kswz(row);
  return *(const bf16x8_t*)(img + (ks >> 1) * ROWS * 128 + row * 128 + cc * 16);
}
// rows 32*s + 8g + {0..7} of the image x 16 columns starting at 16*nb, transposed into a B operand
template <int HDPV>
__device__ __forceinline__ bf16x8_t timg_frag(const char* img, int s, int nb, int lane) {
  constexpr int PC = HDPV / 8;
  constexpr int PITCH = HDPV * 2;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int kr = 32 * s + 8 * g + q;
  const int c = ((2 * nb) + (p >> 1)) ^ mnswz<PC>(kr);
  const char* a0 = img + kr * PITCH + c * 16 + (p & 1) * 8;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lptr_t)a0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(lptr_t)(a0 + 4 * PITCH));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// The same B operand (rows 32*s + 8g + {0..7} x 16 columns starting at 16*nb, transposed) read out of a K-MAJOR
// image, so a tile that is needed in both orientations is staged once.  The K-major swizzle leaves a 2-way bank
// conflict on these reads (rows r and r+2 of a lane group share banks), which costs less than a second image:
// half the LDS-DMA instructions and half the LDS bytes per step.
template <int ROWS>
__device__ __forceinline__ bf16x8_t timg_frag_k(const char* img, int s, int nb, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int r0 = 32 * s + 8 * g + q, r1 = r0 + 4;
  const int c = 2 * (nb & 3) + (p >> 1);
  const char* base = img + (nb >> 2) * ROWS * 128 + (p & 1) * 8;
  const char* a0 = base + r0 * 128 + ((c ^ kswz(r0)) * 16);
  const char* a1 = base + r1 * 128 + ((c ^ kswz(r1)) * 16);
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lptr_t)a0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lptr_t)a1);
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// inline-asm variants: hipcc orders the ds_read_tr builtin behind every outstanding LDS-DMA (s_waitcnt vmcnt(0)),
// which would serialise a prefetch of the next tile behind these reads.  The caller must run lds_wait_all() and
// tie() the fragments before using them.
__device__ __forceinline__ bf16x8_t tr_pair_async(const char* a0, const char* a1) {
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"((uint32_t)(uintptr_t)(lptr_t)a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"((uint32_t)(uintptr_t)(lptr_t)a1));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
template <int ROWS>
__device__ __forceinline__ bf16x8_t timg_frag_k_async(const char* img, int s, int nb, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int r0 = 32 * s + 8 * g + q, r1 = r0 + 4;
  const int c = 2 * (nb & 3) + (p >> 1);
  const char* base = img + (nb >> 2) * ROWS * 128 + (p & 1) * 8;
  return tr_pair_async(base + r0 * 128 + ((c ^ kswz(r0)) * 16), base + r1 * 128 + ((c ^ kswz(r1)) * 16));
}
template <int HDPV>
__device__ __forceinline__ bf16x8_t timg_frag_async(const char* img, int s, int nb, int lane) {
  constexpr int PC = HDPV / 8;
  constexpr int PITCH = HDPV * 2;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int kr = 32 * s + 8 * g + q;
  const int c = ((2 * nb) + (p >> 1)) ^ mnswz<PC>(kr);
  const char* a0 = img + kr * PITCH + c * 16 + (p & 1) * 8;
  return tr_pair_async(a0, a0 + 4 * PITCH);
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tie(bf16x8_t& f) { asm volatile("" : "+v"(f)); }

// ---- fast-path tile loads ---------------------------------------------------------------------------------------
// A tile that lies fully inside its tensor and is not the tensor's last needs no row clamp, and its columns beyond
// head_dim may hold whatever follows in memory (the neighbouring head or the next row: finite activations that only
// ever meet the zero-padded register operand or land in output columns that are never stored).  Its per-lane byte
// offsets relative to the tile's first row are then constants of the kernel and only a wave-uniform base moves from
// tile to tile: one LDS-DMA instruction per 1-KiB piece, no per-lane address arithmetic (the general loaders above
// spend ~20 vector instructions per piece on row clamps, zero-page selects and 64-bit multiplies).  The last tile of
// a tensor always takes the general loader (rows beyond the end, and no read past the allocation).
__device__ __forceinline__ const char* uniform_ptr(const void* p) {  // a wave-uniform pointer the compiler cannot prove uniform
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  // SGPRs written by a VALU instruction (v_readfirstlane) need five wait states before a VMEM instruction reads them,
  // and the LDS-DMA below is inline asm the hazard recogniser does not see: pass the pair through a scalar move, whose
  // result carries no such hazard
  uint64_t r;
  asm volatile("s_mov_b64 %0, %1" : "=s"(r) : "s"(((uint64_t)hi << 32) | lo));
  return (const char*)(uintptr_t)r;
}
__device__ __forceinline__ void glds16_sb(const char* base, uint32_t off, uint32_t lds_piece) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_piece) : "memory", "m0");
}
template <int ROWS, int HDPV, int NW = 4>
struct KImgFast {
  static constexpr int NCH = HDPV / 64, IPW = ROWS / 8 / NW;
  uint32_t off[NCH * IPW];
  __device__ __forceinline__ void init(int64_t ld, int wave, int lane) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        const int r = (wave * IPW + i) * 8 + (lane >> 3);
        const int cc = (lane & 7) ^ kswz(r);
        off[c * IPW + i] = (uint32_t)((r * ld + c * 64 + cc * 8) * 2);
      }
  }
  __device__ __forceinline__ void issue(char* lds, const unsigned short* tile_base, int wave) const {
    const uint32_t l0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)lds);
    const char* tb = uniform_ptr(tile_base);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < IPW; ++i)
        glds16_sb(tb, off[c * IPW + i], l0 + c * ROWS * 128 + (wave * IPW + i) * 1024);
  }
};
template <int ROWS, int HDPV, int NW = 4>
struct MnImgFast {
  static constexpr int PC = HDPV / 8, RPI = 64 / PC, IPW = ROWS * HDPV * 2 / 1024 / NW;
  uint32_t off[IPW];
  __device__ __forceinline__ void init(int64_t ld, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int kr = (wave * IPW + i) * RPI + lane / PC;
      const int c = (lane % PC) ^ mnswz<PC>(kr);
      off[i] = (uint32_t)((kr * ld + c * 8) * 2);
    }
  }
  __device__ __forceinline__ void issue(char* lds, const unsigned short* tile_base, int wave) const {
    const uint32_t l0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)lds);
    const char* tb = uniform_ptr(tile_base);
#pragma unroll
    for (int i = 0; i < IPW; ++i) glds16_sb(tb, off[i], l0 + (wave * IPW + i) * 1024);
  }
};
// number of leading tiles of `rows_per_tile` rows that are fully inside a tensor of n rows and are not its last tile
__device__ __forceinline__ int fast_tiles(int n, int rows_per_tile) { return (n - 1) / rows_per_tile; }

// ---- wave-private output staging: 16 rows x HDPV columns, accumulator layout -> 16-byte row stores --------------
// o[nb][e] is element (row 4g + e, column 16 nb + r) of the wave's 16-row tile.  Stored straight from the registers
// that is one 2-byte store per lane and element (32 branches + stores per lane for 128 columns, 32-byte segments);
// through LDS the tile goes out as whole 16-byte chunks of rows.
// dst8 (optional): the same rows also as e4m3 bytes of the bf16-rounded values times scale8 (delayed per-tensor scale,
// CaAttnDesc.O8) with the lane's running max |value| in amx.
template <int HDPV>
__device__ __forceinline__ void store_tile16(char* st, const f32x4_t (&o)[HDPV / 16], const float (&rs)[4],
                                             unsigned short* dst, int64_t ld, int row0, int nrows, int hd, int lane,
                                             unsigned char* dst8 = nullptr, float scale8 = 1.f, float* amx = nullptr) {
  constexpr int PITCH = HDPV * 2 + 16;  // bytes: consecutive rows start 4 banks apart
  const int g = lane >> 4, r = lane & 15;
#pragma unroll
  for (int nb = 0; nb < HDPV / 16; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *(unsigned short*)(st + (4 * g + e) * PITCH + (16 * nb + r) * 2) = f2bf(o[nb][e] * rs[e]);
  __builtin_amdgcn_wave_barrier();
  constexpr int CPR = HDPV / 8;  // 16-byte chunks per row
#pragma unroll
  for (int i = 0; i < 16 * CPR / 64; ++i) {
    const int idx = i * 64 + lane;
    const int row = idx / CPR, ch = idx % CPR;
    const int q = row0 + row;
    if (q < nrows && ch * 8 < hd) {
      const u16x8_t v = *(const u16x8_t*)(st + row * PITCH + ch * 16);
      *(u16x8_t*)(dst + (int64_t)q * ld + ch * 8) = v;
      if (dst8) {  // (wave-uniform)
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = bf2f(v[e]);
          *amx = fmaxf(*amx, fabsf(f));
          t[e] = fminf(fmaxf(f * scale8, -448.0f), 448.0f);
        }
        unsigned int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], w1, true);
        *(uint2*)(dst8 + (int64_t)q * ld + ch * 8) = make_uint2(w0, w1);
      }
    }
  }
}
constexpr int attn_stage_bytes(int hdpv) { return 4 * 16 * (hdpv * 2 + 16); }

// row of a 32-row step that lane-row r of block bb must read so that lane group g ends up holding the
// contraction indices 8g .. 8g+7 (bb = 0: +0..3, bb = 1: +4..7)
__device__ __forceinline__ int rowperm(int bb, int r) { return 8 * (r >> 2) + 4 * bb + (r & 3); }

__device__ __forceinline__ bf16x8_t pack8(const float (&v)[8]) {
  s16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(v[e]);
  return __builtin_bit_cast(bf16x8_t, o);
}
__device__ __forceinline__ bf16x8_t load_rowfrag(const unsigned short* base, int64_t ld, int row, int ks,
                                                 int lane, int hd) {
  const int dim = 32 * ks + 8 * (lane >> 4);
  if (dim < hd) return *(const bf16x8_t*)(base + (int64_t)row * ld + dim);
  s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(bf16x8_t, z);
}

struct AttnArgs {
  const unsigned short *Q, *K, *V;
  int64_t ldq, ldk, ldv, sqb, skb, svb;  // row strides and per-batch strides (elements)
  unsigned short* O;                     // forward output / backward: saved O is not needed (Dq given)
  int64_t ldo, sob;
  unsigned char* O8;                     // forward, wide kernels: the output also as e4m3 (CaAttnDesc.O8) or null
  const float* o8_scale;
  unsigned int* o8_amax;
  float* lse;                            // [B, H, Tqp]
  const int32_t* klen;                   // [B] or null
  int B, H, Tq, Tk, hd, Tqp, causal;
  float scale;
  // backward
  const unsigned short* dO;
  int64_t lddo, sdob;
  const float* Dq;                       // [B, H, Tqp] rowsum(dO * O)
  unsigned short *dQ, *dK, *dV;
  int64_t lddq, lddk, lddv, sdqb, sdkb, sdvb;
  // dropout on the attention probabilities (training): keep decision of (b, h, q, k) from (seed, flat index)
  float drop_p;
  uint64_t drop_seed;
  // small-query kernel only: the keys of one (clip, head) dealt to nsplit workgroups (CaAttnDesc.split_ws): partial
  // (max, normaliser, unnormalised output) slabs + one arrival counter per (clip, head); nsplit <= 1 = off
  int nsplit;
  CaKeySplit split;         // (nsplit > 1: how the grid's workgroups map to (clip, head) and part)
  float* split_slab;        // [B * H][nsplit][16][SPLIT_ROW]
  unsigned int* split_cnt;  // [B * H], zero between launches
  // greedy decoding with the query projection inside the kernel (attn_fwd_smallq_kernel<.., true>, ca_decode_attn_qproj):
  // q[b, h, :] = (LayerNorm(x[b]) Wq[h*hd : (h+1)*hd, :]^T + bq) - one query per clip
  const unsigned short* qp_x;   // [B, qp_d] residual-stream rows
  int64_t qp_ldx;
  const float *qp_gamma, *qp_beta, *qp_bias;
  const unsigned short* qp_W;   // [H*hd, qp_d] row-major
  int64_t qp_ldw;
  int qp_d;
  float qp_eps;
};
// Flat index of probability (b, h, q, key): rows are padded to a multiple of 4 keys so that 4 consecutive keys from a
// multiple of 4 share one hash (ca_dropout_keep4); the backward kernels regenerate the forward's decisions from it.
__device__ __forceinline__ uint64_t attn_drop_index(const AttnArgs& a, int b, int h, int q, int key) {
  const uint64_t tkp = (uint64_t)((a.Tk + 3) & ~3);
  return (((uint64_t)b * a.H + h) * (uint64_t)a.Tq + (uint64_t)q) * tkp + (uint64_t)key;
}

#define NEG_INF (-__builtin_inff())
#define LOG2E 1.44269504088896340736f

// XCD-aware placement.  Workgroups are dealt round-robin over the 8 XCDs by linear id, and each XCD has
// its own L2.  All tiles of one (batch, head) stream the same K/V (or Q/dO) panels, so they are placed
// on ONE XCD: the i-th workgroup of XCD x works on tile (i % ntile) of pair (i / ntile) * 8 + x.  With the
// plain (tile, head, batch) grid every XCD fetched every panel from HBM (6.5x the algorithmic bytes in
// the PMC counters); placement only affects speed.
__device__ __forceinline__ bool attn_tile_of_block(int ntile, int H, int B, int& tile, int& h, int& b) {
  const int L = blockIdx.x;
  const int x = L & 7, i = L >> 3;
  const int pair = (i / ntile) * 8 + x;
  tile = i % ntile;
  if (pair >= H * B) return false;
  h = pair % H;
  b = pair / H;
  return true;
}
static inline unsigned attn_grid(int ntile, int H, int B) { return (unsigned)(((H * B + 7) / 8) * 8 * ntile); }

// ---- forward ----------------------------------------------------------------------------------------
// One pass over the key tiles with a running maximum (online softmax).  Scores are kept in log2 units
// (s * scale * log2 e), so a probability costs one FMA and one v_exp_f32.  The running maximum of a
// query is shared by the four lane groups that hold its keys (two xor-shuffles per tile); the output
// accumulators, whose rows are queries 4g+e, pick up their rescale factor from the lane that owns the
// query (one shuffle per row) and are only touched when some maximum in the wave moved.  Tiles that lie
// fully inside the valid key range take a path without any per-element masking.
#define NEG_BIG (-1.0e30f)
// 1024 workgroups at the path's shape (8 query tiles x 128 heads): at 4 waves per SIMD they are all resident at
// once; at 3 (148 VGPRs) a second round runs one third full.
template <int HDPV, bool DROP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = HDPV / 32, NNB = HDPV / 16;
  char* Kimg = smem;                  // K-major image of the key tile
  char* Vimg = smem + 64 * HDPV * 2;  // MN-major image of the value tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  int tile, h, b;
  if (!attn_tile_of_block((a.Tq + 63) / 64, a.H, a.B, tile, h, b)) return;
  const int hd = a.hd;
  const unsigned short* Q = a.Q + b * a.sqb + h * hd;
  const unsigned short* K = a.K + b * a.skb + h * hd;
  const unsigned short* V = a.V + b * a.svb + h * hd;
  const int q0 = tile * 64 + wave * 16;
  const int qi = q0 + r;  // this lane's query (column of the transposed score tile)
  const int qrow = qi < a.Tq ? qi : a.Tq - 1;
  int kl = a.Tk;
  if (a.klen) kl = a.klen[b] < kl ? a.klen[b] : kl;
  // all HDPV/32 k-steps and HDPV/16 column blocks are computed: dims >= hd are zero in every LDS image
  bf16x8_t qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) qf[ks] = load_rowfrag(Q, a.ldq, qrow, ks, lane, hd);
  int ntile = (kl + 63) / 64;
  if (a.causal && ntile > tile + 1) ntile = tile + 1;  // keys beyond the tile's last query are masked
  const float c2 = a.scale * LOG2E;

  float m = NEG_BIG, l = 0.f;  // running max (log2 units, same in the 4 lanes of a query) / this lane's partial sum
  f32x4_t o[NNB];
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) o[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  KImgFast<64, HDPV> kfast;
  MnImgFast<64, HDPV> vfast;
  kfast.init(a.ldk, wave, lane);
  vfast.init(a.ldv, wave, lane);
  const int nfast = fast_tiles(a.Tk, 64);
  for (int kt = 0; kt < ntile; ++kt) {
    if (kt < nfast) {
      kfast.issue(Kimg, K + (int64_t)kt * 64 * a.ldk, wave);
      vfast.issue(Vimg, V + (int64_t)kt * 64 * a.ldv, wave);
    } else {
      load_kmajor_image<64, HDPV>(Kimg, K, a.ldk, kt * 64, a.Tk, hd, wave, lane);
      load_mnmajor_image<64, HDPV>(Vimg, V, a.ldv, kt * 64, a.Tk, hd, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // transposed scores of the 64 keys: block (s, bb) holds keys 32 s + 8 g + 4 bb + {0..3} of query qi
    f32x4_t sc[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      const int row = 32 * (blk >> 1) + rowperm(blk & 1, r);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kimg_frag<64>(Kimg, row, ks, lane), qf[ks], acc, 0, 0, 0);
      sc[blk] = acc;
    }
    const bool full = (kt * 64 + 64 <= kl) && !a.causal;  // uniform: no element of this tile is masked
    if (!full) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kt * 64 + 32 * (blk >> 1) + 8 * g + 4 * (blk & 1) + e;
          const bool ok = key < kl && (!a.causal || key <= qi);
          sc[blk][e] = ok ? sc[blk][e] : NEG_BIG;
        }
    }
    float tmax = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
    for (int blk = 1; blk < 4; ++blk)
      tmax = fmaxf(tmax, fmaxf(fmaxf(sc[blk][0], sc[blk][1]), fmaxf(sc[blk][2], sc[blk][3])));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax * c2);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float p[16];
    float sum = 0.f;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv = __builtin_amdgcn_exp2f(fmaf(sc[blk][e], c2, -m_new));
        if (!full) pv = sc[blk][e] > 0.5f * NEG_BIG ? pv : 0.f;
        p[4 * blk + e] = pv;
        sum += pv;  // the normaliser is the sum of ALL probabilities: dropout acts on the normalised ones
      }
    if constexpr (DROP) {
      const float ks = 1.f / (1.f - a.drop_p);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int key0 = kt * 64 + 32 * (blk >> 1) + 8 * g + 4 * (blk & 1);
        const unsigned keep = ca_dropout_keep4(a.drop_seed, attn_drop_index(a, b, h, qrow, key0), a.drop_p);
#pragma unroll
        for (int e = 0; e < 4; ++e) p[4 * blk + e] = ((keep >> e) & 1u) ? p[4 * blk + e] * ks : 0.f;
      }
    }
    l = fmaf(l, alpha, sum);
    if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {  // some query of this wave moved its maximum
      float ar[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) ar[e] = __shfl(alpha, 4 * g + e, 64);
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[nb][e] *= ar[e];
    }
    m = m_new;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float ps[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ps[e] = p[8 * s + e];
      const bf16x8_t pf = pack8(ps);
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb)
        o[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, timg_frag<HDPV>(Vimg, s, nb, lane), o[nb], 0, 0, 0);
    }
    __syncthreads();
  }
  // total of the four lane groups that share a query; lse in natural-log units for the backward
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float lse = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * 0.69314718055994530942f : __builtin_inff();
  if (g == 0 && qi < a.Tq && a.lse) a.lse[((int64_t)b * a.H + h) * a.Tqp + qi] = lse;
  const float inv = l > 0.f ? 1.0f / l : 0.f;  // nothing attended -> zero output
  float ir[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) ir[e] = __shfl(inv, 4 * g + e, 64);
  unsigned short* O = a.O + b * a.sob + h * hd;
  // (every loop iteration ended with a workgroup barrier: the images are free)
  store_tile16<HDPV>(smem + wave * (attn_stage_bytes(HDPV) / 4), o, ir, O, a.ldo, q0, a.Tq, hd, lane);
}

// ---- forward, wide workgroups ---------------------------------------------------------------------------------------
// NW waves x 32 queries per workgroup (two 16-query blocks per wave: every K / V fragment read feeds two MFMAs), key /
// value tiles in a ring of NST image pairs with NST - 1 tiles in flight and counted waits (the fast path's LDS-DMA is
// inline asm in a fixed order: DPT instructions per wave and tile).  Tiles fully inside the valid keys run a body
// without any per-element compare / select (written as a run-time `if`, the compiler turns the masks into selects that
// run for every tile: ~190 of ~540 vector instructions per 64 MFMAs); the ragged last tile and causal tiles run the
// masked body in a loop of their own.
// Measured at the XLS-R-2B shape (B 8, H 16, T 499, hd 120; by switching parts of the kernel off): 10 us before the
// first MFMA (launch, the query fragments and the first tile: latency, not bandwidth), 22 us of loop, 5 us of
// output burst.  In the loop the two waves of a SIMD run in lock step behind the per-tile barrier, so per tile and SIMD
// 2 x 64 MFMAs (2048 cycles) and 2 x ~250 vector instructions (~2200 cycles) add up instead of overlapping.  128
// queries x 4 waves x 2 image pairs (two workgroups per CU, which drift apart) runs 9 % faster than the 64-query
// kernel; 256 queries x 8 waves x 4 pairs (one workgroup per CU) measured the same or slower, so only the former is
// instantiated.
template <int N>
using IC = std::integral_constant<int, N>;
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// WPE = workgroups (waves per SIMD) the register budget is held to.  Round 4: the kernel is bound by vector-instruction
// issue with both waves of a SIMD stalled ~35 % of the time (LDS fragment reads, the per-tile barrier, MFMA results), and
// a THIRD resident wave hides most of that: the head_dim <= 64 form holds 154 registers (512 / 3 = 170), so it runs three
// workgroups per CU - whisper-medium encoder forward 136 -> 124 us per layer, whisper-large-turbo 166 -> 148
// (tools/archive/exp_attn_env.sh, interleaved).  Measured and dropped on the way: forming the scores of tile kt + 1 under the
// softmax of tile kt inside one wave (a third image pair and a second score set, 230 registers: 5 % SLOWER - the
// co-resident waves already overlap the two pipes, what is short is issue slots), and four waves per SIMD at 128
// registers (27 spilled dwords: 183 us).
template <int HDPV, bool DROP, int NW, int NST, int WPE = 2>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn_fwd_wide_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NQ = 2, NKS = HDPV / 32, NNB = HDPV / 16;
  constexpr int IMG = 64 * HDPV * 2, PAIR = 2 * IMG;  // a tile = K-major image of the keys + MN-major image of the values
  constexpr int QPB = NW * 16 * NQ;                   // queries per workgroup
  constexpr int DPT = (HDPV / 64) * (64 / 8 / NW) + 64 * HDPV * 2 / 1024 / NW;  // LDS-DMA instructions per wave and tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  int tile, h, b;
  if (!attn_tile_of_block((a.Tq + QPB - 1) / QPB, a.H, a.B, tile, h, b)) return;
  const int hd = a.hd;
  const unsigned short* Q = a.Q + b * a.sqb + h * hd;
  const unsigned short* K = a.K + b * a.skb + h * hd;
  const unsigned short* V = a.V + b * a.svb + h * hd;
  const int q0 = tile * QPB + wave * 16 * NQ;
  int qi[NQ], qrow[NQ];  // this lane's queries (columns of the transposed score tiles)
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    qi[j] = q0 + 16 * j + r;
    qrow[j] = qi[j] < a.Tq ? qi[j] : a.Tq - 1;
  }
  int kl = a.Tk;
  if (a.klen) kl = a.klen[b] < kl ? a.klen[b] : kl;
  int ntile = (kl + 63) / 64;
  if (a.causal) {  // keys beyond the workgroup's last query are masked
    const int last = (tile * QPB + QPB - 1) / 64 + 1;
    ntile = ntile < last ? ntile : last;
  }
  KImgFast<64, HDPV, NW> kfast;
  MnImgFast<64, HDPV, NW> vfast;
  kfast.init(a.ldk, wave, lane);
  vfast.init(a.ldv, wave, lane);
  const int nfast = fast_tiles(a.Tk, 64);
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* img = smem + stage * PAIR;
    if (kt < nfast) {
      kfast.issue(img, K + (int64_t)kt * 64 * a.ldk, wave);
      vfast.issue(img + IMG, V + (int64_t)kt * 64 * a.ldv, wave);
    } else {
      load_kmajor_image<64, HDPV, NW>(img, K, a.ldk, kt * 64, a.Tk, hd, wave, lane);
      load_mnmajor_image<64, HDPV, NW>(img + IMG, V, a.ldv, kt * 64, a.Tk, hd, wave, lane);
    }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < ntile) issue(t, t);
  // all HDPV/32 k-steps and HDPV/16 column blocks are computed: dims >= hd are zero in every LDS image
  bf16x8_t qf[NQ][NKS];
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[j][ks] = load_rowfrag(Q, a.ldq, qrow[j], ks, lane, hd);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the query fragments are younger than the first tiles)
  const float c2 = a.scale * LOG2E;

  float m[NQ], l[NQ];  // running max (log2 units, same in the 4 lanes of a query) / this lane's partial sum
  f32x4_t o[NQ][NNB];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    m[j] = NEG_BIG;
    l[j] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) o[j][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  // LDS address of this lane's value fragments (column block nb, rows 8g + q of half 0) inside value image 0; the
  // half, the +4 rows of the second read and nothing else are immediates
  uint32_t voff[NNB];
  {
    constexpr int PC = HDPV / 8;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {
      const int c = ((2 * nb) + (p4 >> 1)) ^ mnswz<PC>(8 * g + q4);
      voff[nb] = (uint32_t)(uintptr_t)(lptr_t)smem + IMG + (8 * g + q4) * (HDPV * 2) + c * 16 + (p4 & 1) * 8;
    }
  }
  // transposed scores of the 64 keys of the tile in image pair `stage`: block (s, bb) holds keys 32 s + 8 g + 4 bb +
  // {0..3} of query qi
  auto qk_tile = [&](int stage, f32x4_t (&sc)[NQ][4], auto b0_c, auto b1_c) __attribute__((always_inline)) {
    const char* Kimg = smem + stage * PAIR;
#pragma unroll
    for (int blk = decltype(b0_c)::value; blk < decltype(b1_c)::value; ++blk) {
      const int row = 32 * (blk >> 1) + rowperm(blk & 1, r);
#pragma unroll
      for (int j = 0; j < NQ; ++j) sc[j][blk] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t kfr = kimg_frag<64>(Kimg, row, ks, lane);
#pragma unroll
        for (int j = 0; j < NQ; ++j) sc[j][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[j][ks], sc[j][blk], 0, 0, 0);
      }
    }
  };
  // softmax of the tile's scores (consumed) and the P V product into o
  // softmax of query block j of the tile's scores (consumed): probabilities as the packed A operands pf[j][0..1]
  auto soft_one = [&](auto full_c, int kt, f32x4_t (&sc)[NQ][4], auto j_c, bf16x8_t (&pf)[NQ][2]) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int j = decltype(j_c)::value;
    {
      if constexpr (!FULL) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = kt * 64 + 32 * (blk >> 1) + 8 * g + 4 * (blk & 1) + e;
            const bool ok = key < kl && (!a.causal || key <= qi[j]);
            sc[j][blk][e] = ok ? sc[j][blk][e] : NEG_BIG;
          }
      }
      float p[16];
      float alpha;
      float m_new;
      if constexpr (DROP) {
        // scores in log2 units first: the products are canonical values, so the maximum below compiles to plain
        // v_max3_f32 (fmaxf straight on MFMA outputs makes the compiler canonicalise every operand: v_max x, x), and a
        // probability is one subtraction and one v_exp_f32.  (A hand-written v_max3_f32 on the accumulators is not an
        // option: the hazard recogniser does not see inline asm, and the MFMA -> VALU read then returns stale registers.)
        float t[16];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int e = 0; e < 4; ++e) t[4 * blk + e] = sc[j][blk][e] * c2;
        float tmax = fmaxf(fmaxf(t[0], t[1]), t[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) tmax = fmaxf(fmaxf(tmax, t[i]), t[i + 1]);
        tmax = fmaxf(tmax, t[15]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        m_new = fmaxf(m[j], tmax);
        alpha = __builtin_amdgcn_exp2f(m[j] - m_new);
        float sum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float pv = __builtin_amdgcn_exp2f(t[4 * blk + e] - m_new);
            if constexpr (!FULL) pv = sc[j][blk][e] > 0.5f * NEG_BIG ? pv : 0.f;
            p[4 * blk + e] = pv;
            sum += pv;  // the normaliser is the sum of ALL probabilities: dropout acts on the normalised ones
          }
        const float ks = 1.f / (1.f - a.drop_p);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
          const int key0 = kt * 64 + 32 * (blk >> 1) + 8 * g + 4 * (blk & 1);
          const unsigned keep = ca_dropout_keep4(a.drop_seed, attn_drop_index(a, b, h, qrow[j], key0), a.drop_p);
#pragma unroll
          for (int e = 0; e < 4; ++e) p[4 * blk + e] = ((keep >> e) & 1u) ? p[4 * blk + e] * ks : 0.f;
        }
        l[j] = fmaf(l[j], alpha, sum);
      } else {
        // The kernels are bound by vector-instruction issue (~250 per tile and wave against 32 MFMAs), so
        // the softmax is kept short: the maximum of the RAW scores (median of (x, y, +inf) = max without the
        // canonicalising move the compiler puts in front of fmaxf on MFMA outputs), scale and shift as ONE fma in front
        // of the exponential.  (The row sums as one more column of the P V product - P . 1 on the matrix pipe instead
        // of an addition per probability - were measured too: 4.5 % faster at T = 1500, but sums of bf16-rounded
        // probabilities took the 24-layer logits from 3.7e-2 to 5.2e-2 of the fp32 reference; the fp32 sums stay.)
        const float inf = __builtin_inff();
        float tmax = __builtin_amdgcn_fmed3f(sc[j][0][0], sc[j][0][1], inf);
#pragma unroll
        for (int i = 2; i < 16; ++i) tmax = __builtin_amdgcn_fmed3f(tmax, sc[j][i >> 2][i & 3], inf);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        m_new = fmaxf(m[j], tmax * c2);
        alpha = __builtin_amdgcn_exp2f(m[j] - m_new);
        const float nm = -m_new;
        float sum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float pv = __builtin_amdgcn_exp2f(fmaf(sc[j][blk][e], c2, nm));
            if constexpr (!FULL) pv = sc[j][blk][e] > 0.5f * NEG_BIG ? pv : 0.f;
            p[4 * blk + e] = pv;
            sum += pv;
          }
        l[j] = fmaf(l[j], alpha, sum);
      }
      if (__builtin_amdgcn_ballot_w64(m_new > m[j]) != 0) {  // some query of this block moved its maximum
        float ar[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ar[e] = __shfl(alpha, 4 * g + e, 64);
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[j][nb][e] *= ar[e];
      }
      m[j] = m_new;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float ps[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ps[e] = p[8 * s + e];
        pf[j][s] = pack8(ps);
      }
    }
  };
  // the P V product of the tile in image pair `stage` into o
  auto pv_tile = [&](int stage, bf16x8_t (&pf)[NQ][2]) __attribute__((always_inline)) {
    // value fragments in batches of four column blocks; the next batch is in flight while the current one is used
    // (inline-asm reads: the builtin would be ordered behind the LDS-DMA of the tiles in flight)
    constexpr int NBATCH = 2 * NNB / 4, PITCH = HDPV * 2;
    const uint32_t sbase = (uint32_t)(stage * PAIR);
    bf16x8_t fa[4], fb[4];
    auto read_batch = [&](auto bt_c, bf16x8_t (&f)[4]) {
      constexpr int BT = decltype(bt_c)::value;
      constexpr int S = BT / (NNB / 4), NB0 = (BT % (NNB / 4)) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s16x4_t lo, hi;
        const uint32_t ad = voff[NB0 + i] + sbase;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad), "n"(S * 32 * PITCH));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad), "n"(S * 32 * PITCH + 4 * PITCH));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        f[i] = __builtin_bit_cast(bf16x8_t, v);
      }
    };
    auto step = [&](auto bt_c, bf16x8_t (&cur)[4], bf16x8_t (&nxt)[4]) {
      constexpr int BT = decltype(bt_c)::value;
      constexpr int S = BT / (NNB / 4), NB0 = (BT % (NNB / 4)) * 4;
      lds_wait_all();
#pragma unroll
      for (int i = 0; i < 4; ++i) tie(cur[i]);
      if constexpr (BT + 1 < NBATCH) read_batch(std::integral_constant<int, BT + 1>{}, nxt);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NQ; ++j)
          o[j][NB0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j][S], cur[i], o[j][NB0 + i], 0, 0, 0);
    };
    read_batch(std::integral_constant<int, 0>{}, fa);
    step(std::integral_constant<int, 0>{}, fa, fb);
    step(std::integral_constant<int, 1>{}, fb, fa);
    if constexpr (NBATCH > 2) {
      step(std::integral_constant<int, 2>{}, fa, fb);
      step(std::integral_constant<int, 3>{}, fb, fa);
    }
  };
  auto soft_pv = [&](auto full_c, int kt, int stage, f32x4_t (&sc)[NQ][4]) __attribute__((always_inline)) {
    bf16x8_t pf[NQ][2];
    soft_one(full_c, kt, sc, std::integral_constant<int, 0>{}, pf);
    soft_one(full_c, kt, sc, std::integral_constant<int, 1>{}, pf);
    static_assert(NQ == 2, "two query blocks per wave");
    pv_tile(stage, pf);
  };
  const int nfull = a.causal ? 0 : (kl / 64 < ntile ? kl / 64 : ntile);
  int stage = 0;
  {
    // tile kt lives in image pair kt % NST; tiles kt + 1 .. kt + NST - 2 stay in flight across the wait for tile kt, and
    // the request for tile kt + NST - 1 goes out right after the barrier that frees its pair (one barrier per tile)
    auto run = [&](auto full_c, int kt0, int kt1) __attribute__((always_inline)) {
      for (int kt = kt0; kt < kt1; ++kt) {
        const int ahead = ntile - 1 - kt;
        if (NST >= 4 && ahead >= 2)
          wait_vm<2 * DPT>();
        else if (NST >= 3 && ahead >= 1)
          wait_vm<DPT>();
        else
          wait_vm<0>();
        __syncthreads();
        if (kt + NST - 1 < ntile) issue(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);
        f32x4_t sc[NQ][4];
        qk_tile(stage, sc, IC<0>{}, IC<4>{});
        soft_pv(full_c, kt, stage, sc);
        stage = stage + 1 == NST ? 0 : stage + 1;
      }
    };
    run(std::true_type{}, 0, nfull);
    run(std::false_type{}, nfull, ntile);
  }
  __syncthreads();  // every wave is done with the images: the LDS becomes output staging
  unsigned short* O = a.O + b * a.sob + h * hd;
  unsigned char* O8 = a.O8 ? a.O8 + b * a.sob + h * hd : nullptr;
  const float s8 = a.O8 ? a.o8_scale[0] : 1.f;
  float amx = 0.f;
  char* st = smem + wave * (16 * (HDPV * 2 + 16));
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    // total of the four lane groups that share a query; lse in natural-log units for the backward
    float lt = l[j];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float lse = lt > 0.f ? (m[j] + __builtin_amdgcn_logf(lt)) * 0.69314718055994530942f : __builtin_inff();
    if (g == 0 && qi[j] < a.Tq && a.lse) a.lse[((int64_t)b * a.H + h) * a.Tqp + qi[j]] = lse;
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;  // nothing attended -> zero output
    float ir[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ir[e] = __shfl(inv, 4 * g + e, 64);
    if (j) __builtin_amdgcn_wave_barrier();
    store_tile16<HDPV>(st, o[j], ir, O, a.ldo, q0 + 16 * j, a.Tq, hd, lane, O8, s8, &amx);
  }
  if (a.O8 && a.o8_amax) {  // (every lane arrives; a maximum does not depend on the order)
    amx = wave_max(amx);
    if (lane == 0 && amx > 0.f) atomicMax(a.o8_amax + (blockIdx.x & (CA_FP8_AMAX_SLOTS - 1)), __float_as_uint(amx));
  }
}

// Greedy decoding (Tq <= 16 queries per head, hd <= 64): one workgroup per (clip, head), the four waves split the
// keys in tiles of 64 and merge their (m, l, O) at the end.  The kernel is bound by how many bytes a CU keeps in
// flight (one tile at a time per wave streamed 24 GB/s per CU: 16 us for the 1500 cross-attention keys), so each
// wave keeps THREE tiles in flight: the K fragments go straight from global memory into MFMA operand registers
// (a K row is an A-operand row: no LDS image), the V tile through a wave-private ring of three LDS images
// (LDS-DMA + transposed reads).  All vector-memory operations of the loop are inline asm in a fixed order (16 per
// tile), so the waits are counted by hand: vmcnt(32) leaves the two younger tiles in flight.
__device__ __forceinline__ bf16x8_t gload16_async(const void* p) {
  bf16x8_t v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void glds16_async(const void* g, char* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(g), "s"((uint32_t)(uintptr_t)(lptr_t)lds_wave_base)
               : "memory", "m0");
}
// QP (ca_decode_attn_qproj, one query per clip): the wave-0..3 workgroup of (clip, head) first forms its own query -
// LayerNorm of the clip's residual row (the arithmetic of ln_fwd_kernel, chunk by chunk) and the 64 x d slice of the
// query projection (the arithmetic of ca_gemm_skinny_kernel: the four waves take a quarter of K each, one MFMA chain
// per 16 columns, partials added as (p0 + p1) + (p2 + p3), + bias, rounded to bf16) - so the LayerNorm launch, the
// projection launch and the query's trip through HBM disappear from the per-token chain; bit-identical to them.
#define SPLIT_ROW 66  // floats per query of a partial slab: m, l, 64 output columns
// DT: tiles in flight per wave = slots of its V ring.  3 (96 KiB of LDS: one workgroup per CU) where clips x heads is
// at most about two per CU; 2 (64 KiB: two workgroups per CU) for larger batches (round 5: an evaluation batch of 64
// clips x 16 heads is 1024 workgroups - with one per CU they ran in four rounds, each paying its LayerNorm + query
// projection prologue in front of its K|V stream; with two per CU one workgroup's prologue runs under the other's stream).
template <int HDPV, bool QP = false, int DT = 3>
__global__ __launch_bounds__(256) void attn_fwd_smallq_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = HDPV / 32, NNB = HDPV / 16;
  constexpr int IMG = 64 * HDPV * 2;
  constexpr int D = DT;  // tiles in flight per wave
  static_assert(D == 2 || D == 3, "two or three tiles in flight");
  static_assert(HDPV == 64, "16 vector-memory operations per tile are assumed by the counted waits");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  // Key split (a.nsplit > 1): nsplit workgroups per (clip, head) - with B x H below the CU count (8 clips x 16 heads on
  // 256 CUs) half the chip would idle while each workgroup streams its 384 KB of cross-attention K|V at what ONE CU takes
  // in.  The tiles are dealt to 4 x nsplit "waves"; each workgroup leaves its (m, l, O) partial in a slab and the last
  // one to arrive (one counter per clip and head; release / acquire at agent scope around it) merges them in slab order.
  int ns = 1, bh = blockIdx.x, sp = 0, part0 = blockIdx.x;
  if (a.nsplit > 1) ca_key_split_item(a.split, blockIdx.x, bh, sp, ns, part0);
  const int h = bh % a.H, b = bh / a.H;
  const int vw = sp * 4 + wave, nvw = 4 * ns;  // this wave's place among the waves that share the keys
  const int hd = a.hd;
  const unsigned short* Q = QP ? nullptr : a.Q + b * a.sqb + h * hd;
  const unsigned short* K = a.K + b * a.skb + h * hd;
  const unsigned short* V = a.V + b * a.svb + h * hd;
  char* Vring = smem + wave * D * IMG;
  const int qrow = r < a.Tq ? r : a.Tq - 1;
  int kl = a.Tk;
  if (a.klen) kl = a.klen[b] < kl ? a.klen[b] : kl;
  bf16x8_t qf[NKS];
  if constexpr (QP) {
    const int C = a.qp_d, nchunk = C >> 3;
    unsigned short* xs = (unsigned short*)(smem + 4 * D * IMG);  // LayerNorm(x[b]) as bf16 [C <= 2048]
    float* qpart = (float*)(smem + 4 * D * IMG + 4096);          // [4 waves][64] partial dot products
    {  // every wave normalises the row (redundantly: no barrier before the statistics); wave 0 publishes it
      const unsigned short* xr = a.qp_x + (int64_t)b * a.qp_ldx;
      float v[4][8];
      float sx = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
          const u16x8_t u = *(const u16x8_t*)(xr + ch * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[c][e] = bf2f(u[e]);
            sx += v[c][e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
      }
      const float mean = wave_sum(sx) / (float)C;
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dlt = v[c][e] - mean;
            s2 += dlt * dlt;
          }
        }
      }
      const float rstd = rsqrtf(wave_sum(s2) / (float)C + a.qp_eps);
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ch = lane + c * 64;
          if (ch < nchunk) {
            const f32x4_t g0 = *(const f32x4_t*)(a.qp_gamma + ch * 8), g1 = *(const f32x4_t*)(a.qp_gamma + ch * 8 + 4);
            const f32x4_t b0 = *(const f32x4_t*)(a.qp_beta + ch * 8), b1 = *(const f32x4_t*)(a.qp_beta + ch * 8 + 4);
            u16x8_t o8;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              o8[e] = f2bf(ln_apply(v[c][e], mean, rstd, e < 4 ? g0[e] : g1[e - 4], e < 4 ? b0[e] : b1[e - 4]));
            *(u16x8_t*)(xs + ch * 8) = o8;
          }
        }
      }
    }
    __syncthreads();
    // q[h*hd + n] for n < 64: A operand = the weight rows (16 per block), B operand = the normalised row in every lane
    // row; this wave's quarter of the K-steps, eight at a time, as in the weight-streaming GEMM
    const int ksteps = (C + 31) / 32, per = (ksteps + 3) / 4;
    const int ks0 = wave * per, ks1 = (ks0 + per < ksteps) ? ks0 + per : ksteps;
    const bf16x8_t zero8 = __builtin_bit_cast(bf16x8_t, (f32x4_t){0.f, 0.f, 0.f, 0.f});
    f32x4_t qa[NNB];
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) qa[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned short* wrow[NNB];
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {
      const int n = 16 * nb + r;
      wrow[nb] = a.qp_W + (int64_t)(h * hd + (n < hd ? n : hd - 1)) * a.qp_ldw + 8 * g;
    }
    for (int ks = ks0; ks < ks1; ks += 8) {
      bf16x8_t wf[NNB][8], af[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = (ks + u) * 32 + 8 * g;
        const bool ok = ks + u < ks1 && k < C;
        af[u] = ok ? *(const bf16x8_t*)(xs + k) : zero8;
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) wf[nb][u] = ok ? *(const bf16x8_t*)(wrow[nb] + (int64_t)(ks + u) * 32) : zero8;
      }
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
        for (int u = 0; u < 8; ++u) qa[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][u], af[u], qa[nb], 0, 0, 0);
    }
    if (r == 0) {
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) qpart[wave * 64 + 16 * nb + 4 * g + e] = qa[nb][e];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      float qv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = 32 * ks + 8 * g + j;
        float t = (qpart[n] + qpart[64 + n]) + (qpart[128 + n] + qpart[192 + n]);
        t = t * 1.0f + (n < hd ? a.qp_bias[h * hd + n] : 0.f);
        qv[j] = n < hd ? bf2f(f2bf(t)) : 0.f;  // (the bf16 the projection would have stored)
      }
      qf[ks] = pack8(qv);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = load_rowfrag(Q, a.ldq, qrow, ks, lane, hd);
  }
  const int ntile = (kl + 63) / 64;
  const int nw = ntile > vw ? (ntile - vw + nvw - 1) / nvw : 0;  // this wave's tiles: kt = vw + nvw j
  const float c2 = a.scale * LOG2E;
  float m = NEG_BIG, l = 0.f;
  f32x4_t o[NNB];
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) o[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Q / klen have arrived: from here on the counts are the loop's
  bf16x8_t kf[D][4][NKS];
  // 8 K-fragment loads + 8 LDS-DMA pieces of V for tile j of this wave, into slot S
  auto issue = [&](auto slot_c, int j) {
    constexpr int S = decltype(slot_c)::value;
    const int kt = vw + nvw * j;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      int key = kt * 64 + 32 * (blk >> 1) + rowperm(blk & 1, r);
      key = key < a.Tk ? key : a.Tk - 1;  // clamped rows are masked below (key >= kl)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int dim = 32 * ks + 8 * g;
        kf[S][blk][ks] = gload16_async(K + (int64_t)key * a.ldk + (dim < hd ? dim : 0));
      }
    }
    constexpr int PC = HDPV / 8, RPI = 64 / PC;
    char* img = Vring + S * IMG;
#pragma unroll
    for (int i = 0; i < 64 * HDPV * 2 / 1024; ++i) {
      const int kr = i * RPI + lane / PC;
      const int c = (lane % PC) ^ mnswz<PC>(kr);
      const int dim = c * 8;
      const int row = kt * 64 + kr;
      const void* src = (row < a.Tk && dim < hd) ? (const void*)(V + (int64_t)row * a.ldv + dim)
                                                 : (const void*)g_attn_zero_page;
      glds16_async(src, img + i * 1024);
    }
  };
  auto step = [&](auto slot_c, int j) {
    constexpr int S = decltype(slot_c)::value;
    const int kt = vw + nvw * j;
    const int rem = nw - 1 - j;  // younger tiles already issued (at most D - 1)
    if (D > 2 && rem >= 2)
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (rem >= 1)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* Vimg = Vring + S * IMG;
    // the V fragments are asked for first: their LDS latency hides under the score MFMAs
    bf16x8_t vf[2][NNB];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) vf[s][nb] = timg_frag_async<HDPV>(Vimg, s, nb, lane);
    f32x4_t sc[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        tie(kf[S][blk][ks]);
        if (32 * ks + 8 * g >= hd) kf[S][blk][ks] = __builtin_bit_cast(bf16x8_t, (f32x4_t){0.f, 0.f, 0.f, 0.f});
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[S][blk][ks], qf[ks], acc, 0, 0, 0);
      }
      sc[blk] = acc;
    }
    // (tiles wholly inside the valid keys: a body of their own without compare / select - a run-time flag inside the
    // element loops becomes control flow per element)
    auto rest = [&](auto full_c) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    if constexpr (!FULL) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kt * 64 + 32 * (blk >> 1) + 8 * g + 4 * (blk & 1) + e;
          sc[blk][e] = key < kl ? sc[blk][e] : NEG_BIG;
        }
    }
    float tmax = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
    for (int blk = 1; blk < 4; ++blk)
      tmax = fmaxf(tmax, fmaxf(fmaxf(sc[blk][0], sc[blk][1]), fmaxf(sc[blk][2], sc[blk][3])));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax * c2);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float p[16];
    float sum = 0.f;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv = __builtin_amdgcn_exp2f(fmaf(sc[blk][e], c2, -m_new));
        if constexpr (!FULL) pv = sc[blk][e] > 0.5f * NEG_BIG ? pv : 0.f;
        p[4 * blk + e] = pv;
        sum += pv;  // the normaliser is the sum of ALL probabilities: dropout acts on the normalised ones
      }

    l = fmaf(l, alpha, sum);
    float ar[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ar[e] = __shfl(alpha, 4 * g + e, 64);
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[nb][e] *= ar[e];
    m = m_new;
    lds_wait_all();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float ps[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ps[e] = p[8 * s + e];
      const bf16x8_t pf = pack8(ps);
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) {
        tie(vf[s][nb]);
        o[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf[s][nb], o[nb], 0, 0, 0);
      }
    }
    };
    if (kt * 64 + 64 <= kl)
      rest(std::true_type{});
    else
      rest(std::false_type{});
    if (j + D < nw) issue(slot_c, j + D);  // slot S is free again: its registers and its LDS image have been consumed
  };
  if (nw > 0) issue(std::integral_constant<int, 0>{}, 0);
  if (nw > 1) issue(std::integral_constant<int, 1>{}, 1);
  if constexpr (D > 2) {
    if (nw > 2) issue(std::integral_constant<int, 2>{}, 2);
  }
  for (int j = 0; j < nw; j += D) {
    step(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nw) step(std::integral_constant<int, 1>{}, j + 1);
    if constexpr (D > 2) {
      if (j + 2 < nw) step(std::integral_constant<int, 2>{}, j + 2);
    }
  }
  // merge the four waves: (m, l) per query and the output rows, through LDS (the images are dead now)
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  __syncthreads();
  float* cm = (float*)smem;              // [4][16]
  float* cl = cm + 64;                   // [4][16]
  float* co = cl + 64;                   // [4][16][HDPV]
  if (g == 0) {
    cm[wave * 16 + r] = m;
    cl[wave * 16 + r] = l;
  }
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) co[(wave * 16 + 4 * g + e) * HDPV + 16 * nb + r] = o[nb][e];
  __syncthreads();
  if (wave != 0) return;
  unsigned short* O = a.O + b * a.sob + h * hd;
  if (ns > 1) {
    // this workgroup's partial: per query (m, l) and the unnormalised output row
    float* slab = a.split_slab + ((int64_t)part0 + sp) * (16 * SPLIT_ROW);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = 4 * g + e;
      float M = cm[q];
#pragma unroll
      for (int w = 1; w < 4; ++w) M = fmaxf(M, cm[w * 16 + q]);
      float L = 0.f, wgt[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        wgt[w] = __builtin_amdgcn_exp2f(cm[w * 16 + q] - M);
        L = fmaf(cl[w * 16 + q], wgt[w], L);
      }
      if (q < a.Tq) {
        if (r == 0) {
          slab[q * SPLIT_ROW] = M;
          slab[q * SPLIT_ROW + 1] = L;
        }
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) v = fmaf(co[(w * 16 + q) * HDPV + 16 * nb + r], wgt[w], v);
          slab[q * SPLIT_ROW + 2 + 16 * nb + r] = v;
        }
      }
    }
    // publish (plain stores -> drained -> agent-scope release -> ticket); the workgroup that draws the last ticket
    // acquires and merges.  Nobody waits for anybody: no residency assumption.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(a.split_cnt + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != (unsigned)(ns - 1)) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* base = a.split_slab + (int64_t)part0 * (16 * SPLIT_ROW);
    // lane = output column; every slab value of a query is asked for before the first is used (one round trip per
    // query - a decoded token has one query per clip - instead of one per slab and value)
    for (int q = 0; q < a.Tq; ++q) {
      float Mt[CA_ATTN_SPLIT_MAX], Lt[CA_ATTN_SPLIT_MAX], ot[CA_ATTN_SPLIT_MAX];
#pragma unroll
      for (int t = 0; t < CA_ATTN_SPLIT_MAX; ++t) {
        const float* row = base + ((t < ns ? t : 0) * 16 + q) * SPLIT_ROW;
        Mt[t] = row[0];
        Lt[t] = row[1];
        ot[t] = row[2 + lane];
      }
      float M = NEG_BIG;
#pragma unroll
      for (int t = 0; t < CA_ATTN_SPLIT_MAX; ++t) M = fmaxf(M, t < ns ? Mt[t] : NEG_BIG);
      float L = 0.f, v = 0.f;
#pragma unroll
      for (int t = 0; t < CA_ATTN_SPLIT_MAX; ++t) {
        const float wt = t < ns ? __builtin_amdgcn_exp2f(Mt[t] - M) : 0.f;
        L = fmaf(Lt[t], wt, L);
        v = fmaf(ot[t], wt, v);
      }
      const float inv = L > 0.f ? 1.0f / L : 0.f;
      if (lane < hd) O[(int64_t)q * a.ldo + lane] = f2bf(v * inv);
      if (lane == 0 && a.lse)
        a.lse[((int64_t)b * a.H + h) * a.Tqp + q] = L > 0.f ? (M + __builtin_amdgcn_logf(L)) * 0.69314718055994530942f
                                                            : __builtin_inff();
    }
    if (lane == 0) __hip_atomic_store(a.split_cnt + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = 4 * g + e;
    float M = cm[q];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, cm[w * 16 + q]);
    float L = 0.f, wgt[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      wgt[w] = __builtin_amdgcn_exp2f(cm[w * 16 + q] - M);
      L = fmaf(cl[w * 16 + q], wgt[w], L);
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    if (q < a.Tq) {
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) {
        const int n = 16 * nb + r;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v = fmaf(co[(w * 16 + q) * HDPV + n], wgt[w], v);
        if (n < hd) O[(int64_t)q * a.ldo + n] = f2bf(v * inv);
      }
      if (r == 0 && a.lse)
        a.lse[((int64_t)b * a.H + h) * a.Tqp + q] = L > 0.f ? (M + __builtin_amdgcn_logf(L)) * 0.69314718055994530942f
                                                            : __builtin_inff();
    }
  }
}

// ---- backward: D = rowsum(dO * O) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const unsigned short* __restrict__ dO, int64_t lddo,
                                                            int64_t sdob, const unsigned short* __restrict__ O,
                                                            int64_t ldo, int64_t sob, float* __restrict__ Dq, int H,
                                                            int Tq, int Tqp, int hd, int B) {
  // one 16-lane group per (b, h, q)
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  const int64_t total = (int64_t)B * H * Tq;
  if (gid >= total) return;
  const int q = (int)(gid % Tq);
  const int h = (int)((gid / Tq) % H);
  const int b = (int)(gid / ((int64_t)Tq * H));
  const unsigned short* x = dO + b * sdob + (int64_t)q * lddo + h * hd;
  const unsigned short* y = O + b * sob + (int64_t)q * ldo + h * hd;
  float s = 0.f;
  for (int c = sub * 8; c < hd; c += 128) {
    const u16x8_t u = *(const u16x8_t*)(x + c), v = *(const u16x8_t*)(y + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f(u[e]) * bf2f(v[e]);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (sub == 0) Dq[((int64_t)b * H + h) * Tqp + q] = s;
}

// ---- backward: dK, dV (workgroup = 64 keys, loops over the queries) ------------------------------------
template <int HDPV, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = HDPV / 32, NNB = HDPV / 16;
  constexpr int IMG = 32 * HDPV * 2;
  char* Qk = smem;         // K-major image of the 32-query tile: S = Q K^T, and (read transposed) dK += dS^T Q
  char* dOk = smem + IMG;  // K-major image of dO:                dP = dO V^T, and (read transposed) dV += P^T dO
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  int tile, h, b;
  if (!attn_tile_of_block((a.Tk + 63) / 64, a.H, a.B, tile, h, b)) return;
  const int hd = a.hd;
  const unsigned short* Q = a.Q + b * a.sqb + h * hd;
  const unsigned short* K = a.K + b * a.skb + h * hd;
  const unsigned short* V = a.V + b * a.svb + h * hd;
  const unsigned short* dO = a.dO + b * a.sdob + h * hd;
  const float* lse = a.lse + ((int64_t)b * a.H + h) * a.Tqp;
  const float* Dq = a.Dq + ((int64_t)b * a.H + h) * a.Tqp;
  const int k0 = tile * 64 + wave * 16;
  const int key = k0 + r;  // this lane's key (column of the score tile)
  const int krow = key < a.Tk ? key : a.Tk - 1;
  int kl = a.Tk;
  if (a.klen) kl = a.klen[b] < kl ? a.klen[b] : kl;
  // all HDPV/32 k-steps and HDPV/16 column blocks are computed: dims >= hd are zero in every LDS image
  bf16x8_t kf[NKS], vf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    kf[ks] = load_rowfrag(K, a.ldk, krow, ks, lane, hd);
    vf[ks] = load_rowfrag(V, a.ldv, krow, ks, lane, hd);
  }
  f32x4_t dk[NNB], dv[NNB];
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) dk[nb] = dv[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float scale = a.scale;
  const float c2 = scale * LOG2E;
  const bool full_keys = (tile * 64 + 64 <= kl) && !a.causal;
  const int nstep = (a.Tq + 31) / 32;
  const int s0 = a.causal ? (tile * 64) / 32 : 0;  // queries before the tile's first key see none of it
  // Double-buffered: the images and the per-query statistics of step qs+1 are requested right after the barrier
  // of step qs (one barrier per step: every wave has finished reading the other buffer when it arrives).
  KImgFast<32, HDPV> qfast, dofast;
  qfast.init(a.ldq, wave, lane);
  dofast.init(a.lddo, wave, lane);
  const int nfast = fast_tiles(a.Tq, 32);
  auto issue = [&](int qs, int buf) {
    if (qs < nfast) {
      qfast.issue(Qk + buf * 2 * IMG, Q + (int64_t)qs * 32 * a.ldq, wave);
      dofast.issue(dOk + buf * 2 * IMG, dO + (int64_t)qs * 32 * a.lddo, wave);
    } else {
      load_kmajor_image<32, HDPV>(Qk + buf * 2 * IMG, Q, a.ldq, qs * 32, a.Tq, hd, wave, lane);
      load_kmajor_image<32, HDPV>(dOk + buf * 2 * IMG, dO, a.lddo, qs * 32, a.Tq, hd, wave, lane);
    }
  };
  f32x4_t l0, l1, d0, d1;  // statistics of the current step: indices 8g .. 8g+7 are contiguous
  if (s0 < nstep) {
    issue(s0, 0);
    l0 = *(const f32x4_t*)(lse + s0 * 32 + 8 * g);
    l1 = *(const f32x4_t*)(lse + s0 * 32 + 8 * g + 4);
    d0 = *(const f32x4_t*)(Dq + s0 * 32 + 8 * g);
    d1 = *(const f32x4_t*)(Dq + s0 * 32 + 8 * g + 4);
  }
  // One 32-query step; FULL (compile time): nothing in the step is masked - no per-element compare / select.  As a
  // run-time flag inside the element loop the masks became ~30 small basic blocks per step: in-kernel stamps put the
  // softmax / dS section at 1 630 cycles of a 3 430-cycle step (tools/archive/dev_dkv_stamps.py), against 553 + 655 for the 32 MFMAs.
  auto step = [&](auto full_c, int qs) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    const int buf = (qs - s0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4_t nl0 = l0, nl1 = l1, nd0 = d0, nd1 = d1;
    if (qs + 1 < nstep) {
      issue(qs + 1, buf ^ 1);
      nl0 = *(const f32x4_t*)(lse + (qs + 1) * 32 + 8 * g);
      nl1 = *(const f32x4_t*)(lse + (qs + 1) * 32 + 8 * g + 4);
      nd0 = *(const f32x4_t*)(Dq + (qs + 1) * 32 + 8 * g);
      nd1 = *(const f32x4_t*)(Dq + (qs + 1) * 32 + 8 * g + 4);
    }
    const char* Qi = Qk + buf * 2 * IMG;
    const char* dOi = dOk + buf * 2 * IMG;
    f32x4_t sacc[2], pacc[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      sacc[bb] = pacc[bb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const int row = rowperm(bb, r);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        sacc[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kimg_frag<32>(Qi, row, ks, lane), kf[ks], sacc[bb], 0, 0, 0);
        pacc[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kimg_frag<32>(dOi, row, ks, lane), vf[ks], pacc[bb], 0, 0, 0);
      }
    }
    // transposed fragments of the first half of the columns fly while the softmax arithmetic runs
    constexpr int HB = NNB / 2;
    bf16x8_t fo[HB], fq[HB];
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      fo[i] = timg_frag_k_async<32>(dOi, 0, i, lane);
      fq[i] = timg_frag_k_async<32>(Qi, 0, i, lane);
    }
    float p[8], ds[8];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ls2 = (bb == 0 ? l0[e] : l1[e]) * LOG2E;
        const float dq = bb == 0 ? d0[e] : d1[e];
        float pv = __builtin_amdgcn_exp2f(fmaf(sacc[bb][e], c2, -ls2));
        float dpv = pacc[bb][e];  // d/dP of the (dropped) probabilities
        float pdrop = pv;         // the probability as the forward used it against V
        if constexpr (DROP) {
          const int qd = qs * 32 + 8 * g + 4 * bb + e;
          const bool keep = ca_dropout_keep(a.drop_seed, attn_drop_index(a, b, h, qd < a.Tq ? qd : a.Tq - 1, krow), a.drop_p);
          const float ks = 1.f / (1.f - a.drop_p);
          dpv = keep ? dpv * ks : 0.f;
          pdrop = keep ? pv * ks : 0.f;
        }
        float dsv = pv * (dpv - dq);  // (x scale: once, when dK is stored)
        pv = pdrop;
        if constexpr (!FULL) {  // statistics of padded queries are not initialised: select, never multiply by a mask
          const int qi = qs * 32 + 8 * g + 4 * bb + e;
          const bool ok = qi < a.Tq && key < kl && (!a.causal || key <= qi);
          pv = ok ? pv : 0.f;
          dsv = ok ? dsv : 0.f;
        }
        p[4 * bb + e] = pv;
        ds[4 * bb + e] = dsv;
      }
    const bf16x8_t pf = pack8(p), dsf = pack8(ds);
    lds_wait_all();
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      tie(fo[i]);
      tie(fq[i]);
    }
    bf16x8_t go[HB], gq[HB];
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      go[i] = timg_frag_k_async<32>(dOi, 0, HB + i, lane);
      gq[i] = timg_frag_k_async<32>(Qi, 0, HB + i, lane);
    }
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      dv[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, fo[i], dv[i], 0, 0, 0);
      dk[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, fq[i], dk[i], 0, 0, 0);
    }
    lds_wait_all();
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      tie(go[i]);
      tie(gq[i]);
    }
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      dv[HB + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, go[i], dv[HB + i], 0, 0, 0);
      dk[HB + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, gq[i], dk[HB + i], 0, 0, 0);
    }
    l0 = nl0;
    l1 = nl1;
    d0 = nd0;
    d1 = nd1;
  };
  int nfull = full_keys ? a.Tq / 32 : s0;  // steps [s0, nfull) see whole 32-query blocks of valid queries and keys
  nfull = nfull < s0 ? s0 : nfull;
  for (int qs = s0; qs < nfull; ++qs) step(std::true_type{}, qs);
  for (int qs = nfull; qs < nstep; ++qs) step(std::false_type{}, qs);
  unsigned short* dK = a.dK + b * a.sdkb + h * hd;
  unsigned short* dV = a.dV + b * a.sdvb + h * hd;
  __syncthreads();  // the last step's image reads are done in every wave: the LDS becomes output staging
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  char* st = smem + wave * (attn_stage_bytes(HDPV) / 4);
  const float sc4[4] = {scale, scale, scale, scale};  // dS was formed without the softmax scale
  store_tile16<HDPV>(st, dk, sc4, dK, a.lddk, k0, a.Tk, hd, lane);
  __builtin_amdgcn_wave_barrier();
  store_tile16<HDPV>(st, dv, one, dV, a.lddv, k0, a.Tk, hd, lane);
}

// ---- backward: dK, dV, wide waves (workgroup = 4 waves x KB key blocks of 16 = 64 * KB keys) --------------------------
// In the kernel above a wave owns 16 keys, so every Q / dO fragment it reads from LDS feeds ONE MFMA: 80 LDS read
// instructions (48 KiB, the transposed ones with a 2-way bank conflict) per 32 MFMAs and 32-query step - at four
// workgroups per CU the LDS, not the matrix pipe, sets the pace (768 LDS cycles against 512 MFMA cycles per step).
// Here a wave owns KB = 2 key blocks: each fragment feeds two MFMAs, a workgroup covers 128 keys, half as many
// workgroups stream the same Q / dO images.
template <int HDPV, bool DROP, int KB, int WPE = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn_bwd_dkv_wide_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = HDPV / 32, NNB = HDPV / 16;
  constexpr int IMG = 32 * HDPV * 2;
  char* Qk = smem;         // K-major image of the 32-query tile: S = Q K^T, and (read transposed) dK += dS^T Q
  char* dOk = smem + IMG;  // K-major image of dO:                dP = dO V^T, and (read transposed) dV += P^T dO
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  int tile, h, b;
  if (!attn_tile_of_block((a.Tk + 64 * KB - 1) / (64 * KB), a.H, a.B, tile, h, b)) return;
  const int hd = a.hd;
  const unsigned short* Q = a.Q + b * a.sqb + h * hd;
  const unsigned short* K = a.K + b * a.skb + h * hd;
  const unsigned short* V = a.V + b * a.svb + h * hd;
  const unsigned short* dO = a.dO + b * a.sdob + h * hd;
  const float* lse = a.lse + ((int64_t)b * a.H + h) * a.Tqp;
  const float* Dq = a.Dq + ((int64_t)b * a.H + h) * a.Tqp;
  const int k0 = tile * 64 * KB + wave * 16 * KB;  // block kb: keys k0 + 16 kb + r
  int kl = a.Tk;
  if (a.klen) kl = a.klen[b] < kl ? a.klen[b] : kl;
  // all HDPV/32 k-steps and HDPV/16 column blocks are computed: dims >= hd are zero in every LDS image
  bf16x8_t kf[KB][NKS], vf[KB][NKS];
  f32x4_t dk[KB][NNB], dv[KB][NNB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int key = k0 + 16 * kb + r;
    const int krow = key < a.Tk ? key : a.Tk - 1;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[kb][ks] = load_rowfrag(K, a.ldk, krow, ks, lane, hd);
      vf[kb][ks] = load_rowfrag(V, a.ldv, krow, ks, lane, hd);
    }
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) dk[kb][nb] = dv[kb][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const float scale = a.scale;
  const float c2 = scale * LOG2E;
  const bool full_keys = (tile * 64 * KB + 64 * KB <= kl) && !a.causal;
  const int nstep = (a.Tq + 31) / 32;
  const int s0 = a.causal ? (tile * 64 * KB) / 32 : 0;  // queries before the tile's first key see none of it
  // Double-buffered: the images and the per-query statistics of step qs+1 are requested right after the barrier
  // of step qs (one barrier per step: every wave has finished reading the other buffer when it arrives).
  KImgFast<32, HDPV> qfast, dofast;
  qfast.init(a.ldq, wave, lane);
  dofast.init(a.lddo, wave, lane);
  const int nfast = fast_tiles(a.Tq, 32);
  auto issue = [&](int qs, int buf) {
    if (qs < nfast) {
      qfast.issue(Qk + buf * 2 * IMG, Q + (int64_t)qs * 32 * a.ldq, wave);
      dofast.issue(dOk + buf * 2 * IMG, dO + (int64_t)qs * 32 * a.lddo, wave);
    } else {
      load_kmajor_image<32, HDPV>(Qk + buf * 2 * IMG, Q, a.ldq, qs * 32, a.Tq, hd, wave, lane);
      load_kmajor_image<32, HDPV>(dOk + buf * 2 * IMG, dO, a.lddo, qs * 32, a.Tq, hd, wave, lane);
    }
  };
  f32x4_t l0, l1, d0, d1;  // statistics of the current step: indices 8g .. 8g+7 are contiguous
  if (s0 < nstep) {
    issue(s0, 0);
    l0 = *(const f32x4_t*)(lse + s0 * 32 + 8 * g);
    l1 = *(const f32x4_t*)(lse + s0 * 32 + 8 * g + 4);
    d0 = *(const f32x4_t*)(Dq + s0 * 32 + 8 * g);
    d1 = *(const f32x4_t*)(Dq + s0 * 32 + 8 * g + 4);
  }
  // One 32-query step; FULL (compile time): nothing in the step is masked - no per-element compare / select.  As a
  // run-time flag inside the element loop the masks became ~30 small basic blocks per step: in-kernel stamps put the
  // softmax / dS section at 1 630 cycles of a 3 430-cycle step (tools/archive/dev_dkv_stamps.py), against 553 + 655 for the 32 MFMAs.
  auto step = [&](auto full_c, int qs) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    const int buf = (qs - s0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qs + 1 < nstep) issue(qs + 1, buf ^ 1);
    // (this step's statistics: asked for here, used behind the S | dP products - no second set of registers for the next step's)
    l0 = *(const f32x4_t*)(lse + qs * 32 + 8 * g);
    l1 = *(const f32x4_t*)(lse + qs * 32 + 8 * g + 4);
    d0 = *(const f32x4_t*)(Dq + qs * 32 + 8 * g);
    d1 = *(const f32x4_t*)(Dq + qs * 32 + 8 * g + 4);
    const char* Qi = Qk + buf * 2 * IMG;
    const char* dOi = dOk + buf * 2 * IMG;
    f32x4_t sacc[KB][2], pacc[KB][2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) sacc[kb][bb] = pacc[kb][bb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const int row = rowperm(bb, r);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t qf = kimg_frag<32>(Qi, row, ks, lane), dof = kimg_frag<32>(dOi, row, ks, lane);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {  // one fragment read, KB MFMAs
          sacc[kb][bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[kb][ks], sacc[kb][bb], 0, 0, 0);
          pacc[kb][bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vf[kb][ks], pacc[kb][bb], 0, 0, 0);
        }
      }
    }
    // transposed fragments of the first half of the columns fly while the softmax arithmetic runs
    constexpr int HB = NNB / 2;
    bf16x8_t fo[HB], fq[HB];
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      fo[i] = timg_frag_k_async<32>(dOi, 0, i, lane);
      fq[i] = timg_frag_k_async<32>(Qi, 0, i, lane);
    }
    bf16x8_t pf[KB], dsf[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int key = k0 + 16 * kb + r;
      const int krow = key < a.Tk ? key : a.Tk - 1;
      float p[8], ds[8];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ls2 = (bb == 0 ? l0[e] : l1[e]) * LOG2E;
          const float dq = bb == 0 ? d0[e] : d1[e];
          float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][bb][e], c2, -ls2));
          float dpv = pacc[kb][bb][e];  // d/dP of the (dropped) probabilities
          float pdrop = pv;             // the probability as the forward used it against V
          if constexpr (DROP) {
            const int qd = qs * 32 + 8 * g + 4 * bb + e;
            const bool keep = ca_dropout_keep(a.drop_seed, attn_drop_index(a, b, h, qd < a.Tq ? qd : a.Tq - 1, krow), a.drop_p);
            const float ks = 1.f / (1.f - a.drop_p);
            dpv = keep ? dpv * ks : 0.f;
            pdrop = keep ? pv * ks : 0.f;
          }
          float dsv = pv * (dpv - dq);  // (x scale: once, when dK is stored)
          pv = pdrop;
          if constexpr (!FULL) {  // statistics of padded queries are not initialised: select, never multiply by a mask
            const int qi = qs * 32 + 8 * g + 4 * bb + e;
            const bool ok = qi < a.Tq && key < kl && (!a.causal || key <= qi);
            pv = ok ? pv : 0.f;
            dsv = ok ? dsv : 0.f;
          }
          p[4 * bb + e] = pv;
          ds[4 * bb + e] = dsv;
        }
      pf[kb] = pack8(p);
      dsf[kb] = pack8(ds);
    }
    lds_wait_all();
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      tie(fo[i]);
      tie(fq[i]);
    }
    bf16x8_t go[HB], gq[HB];
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      go[i] = timg_frag_k_async<32>(dOi, 0, HB + i, lane);
      gq[i] = timg_frag_k_async<32>(Qi, 0, HB + i, lane);
    }
#pragma unroll
    for (int i = 0; i < HB; ++i)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        dv[kb][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[kb], fo[i], dv[kb][i], 0, 0, 0);
        dk[kb][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf[kb], fq[i], dk[kb][i], 0, 0, 0);
      }
    lds_wait_all();
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      tie(go[i]);
      tie(gq[i]);
    }
#pragma unroll
    for (int i = 0; i < HB; ++i)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        dv[kb][HB + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[kb], go[i], dv[kb][HB + i], 0, 0, 0);
        dk[kb][HB + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf[kb], gq[i], dk[kb][HB + i], 0, 0, 0);
      }
  };
  int nfull = full_keys ? a.Tq / 32 : s0;  // steps [s0, nfull) see whole 32-query blocks of valid queries and keys
  nfull = nfull < s0 ? s0 : nfull;
  for (int qs = s0; qs < nfull; ++qs) step(std::true_type{}, qs);
  for (int qs = nfull; qs < nstep; ++qs) step(std::false_type{}, qs);
  unsigned short* dK = a.dK + b * a.sdkb + h * hd;
  unsigned short* dV = a.dV + b * a.sdvb + h * hd;
  __syncthreads();  // the last step's image reads are done in every wave: the LDS becomes output staging
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  const float sc4[4] = {scale, scale, scale, scale};  // dS was formed without the softmax scale
  char* st = smem + wave * (attn_stage_bytes(HDPV) / 4);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    store_tile16<HDPV>(st, dk[kb], sc4, dK, a.lddk, k0 + 16 * kb, a.Tk, hd, lane);
    __builtin_amdgcn_wave_barrier();
    store_tile16<HDPV>(st, dv[kb], one, dV, a.lddv, k0 + 16 * kb, a.Tk, hd, lane);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- backward: dQ (workgroup = 64 queries, loops over the keys) ------------------------------------------
template <int HDPV, bool DROP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bwd_dq_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = HDPV / 32, NNB = HDPV / 16;
  constexpr int IMG = 64 * HDPV * 2;
  char* Kk = smem;        // K-major image of the key tile: S^T = K Q^T, and (read transposed) dQ += dS K
  char* Vk = smem + IMG;  // K-major image of V:            dP^T = V dO^T
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  int tile, h, b;
  if (!attn_tile_of_block((a.Tq + 63) / 64, a.H, a.B, tile, h, b)) return;
  const int hd = a.hd;
  const unsigned short* Q = a.Q + b * a.sqb + h * hd;
  const unsigned short* K = a.K + b * a.skb + h * hd;
  const unsigned short* V = a.V + b * a.svb + h * hd;
  const unsigned short* dO = a.dO + b * a.sdob + h * hd;
  const int q0 = tile * 64 + wave * 16;
  const int qi = q0 + r;
  const int qrow = qi < a.Tq ? qi : a.Tq - 1;
  int kl = a.Tk;
  if (a.klen) kl = a.klen[b] < kl ? a.klen[b] : kl;
  // all HDPV/32 k-steps and HDPV/16 column blocks are computed: dims >= hd are zero in every LDS image
  bf16x8_t qf[NKS], dof[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    qf[ks] = load_rowfrag(Q, a.ldq, qrow, ks, lane, hd);
    dof[ks] = load_rowfrag(dO, a.lddo, qrow, ks, lane, hd);
  }
  const float lse2 = a.lse[((int64_t)b * a.H + h) * a.Tqp + qrow] * LOG2E;
  const float dq_row = a.Dq[((int64_t)b * a.H + h) * a.Tqp + qrow];
  f32x4_t acc[NNB];
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) acc[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float scale = a.scale;
  const float c2 = scale * LOG2E;
  int ntile = (kl + 63) / 64;
  if (a.causal) {
    const int last = (tile * 64 + 63) / 64 + 1;  // keys beyond the tile's last query are masked
    ntile = ntile < last ? ntile : last;
  }
  KImgFast<64, HDPV> kfast, vfast;
  kfast.init(a.ldk, wave, lane);
  vfast.init(a.ldv, wave, lane);
  const int nfast = fast_tiles(a.Tk, 64);
  for (int kt = 0; kt < ntile; ++kt) {
    if (kt < nfast) {
      kfast.issue(Kk, K + (int64_t)kt * 64 * a.ldk, wave);
      vfast.issue(Vk, V + (int64_t)kt * 64 * a.ldv, wave);
    } else {
      load_kmajor_image<64, HDPV>(Kk, K, a.ldk, kt * 64, a.Tk, hd, wave, lane);
      load_kmajor_image<64, HDPV>(Vk, V, a.ldv, kt * 64, a.Tk, hd, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool full = (kt * 64 + 64 <= kl) && !a.causal;  // uniform: no key of this tile is masked
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float ds[8];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        f32x4_t sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
        const int row = 32 * s + rowperm(bb, r);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
          {
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kimg_frag<64>(Kk, row, ks, lane), qf[ks], sacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kimg_frag<64>(Vk, row, ks, lane), dof[ks], pacc, 0, 0, 0);
          }
        unsigned keep = 0xFu;
        if constexpr (DROP)
          keep = ca_dropout_keep4(a.drop_seed, attn_drop_index(a, b, h, qrow, kt * 64 + 32 * s + 8 * g + 4 * bb), a.drop_p);
        const float kscale = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dpv = DROP ? (((keep >> e) & 1u) ? pacc[e] * kscale : 0.f) : pacc[e];
          float dsv = __builtin_amdgcn_exp2f(fmaf(sacc[e], c2, -lse2)) * (dpv - dq_row);  // (x scale: when dQ is stored)
          if (!full) {
            const int key = kt * 64 + 32 * s + 8 * g + 4 * bb + e;
            const bool ok = key < kl && (!a.causal || key <= qi);
            dsv = ok ? dsv : 0.f;
          }
          ds[4 * bb + e] = dsv;
        }
      }
      const bf16x8_t dsf = pack8(ds);
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, timg_frag_k<64>(Kk, s, nb, lane), acc[nb], 0, 0, 0);
    }
    __syncthreads();
  }
  unsigned short* dQ = a.dQ + b * a.sdqb + h * hd;
  const float sc4[4] = {scale, scale, scale, scale};  // dS was formed without the softmax scale
  // (every loop iteration ended with a workgroup barrier: the images are free)
  store_tile16<HDPV>(smem + wave * (attn_stage_bytes(HDPV) / 4), acc, sc4, dQ, a.lddq, q0, a.Tq, hd, lane);
}

// ---- backward: dQ, wide workgroups (same reasoning and ring as attn_fwd_wide_kernel) -----------------------------------
// NW waves x 32 queries; per 64-key tile a K-major image of K (S^T = K Q^T and, read transposed, dQ += dS K) and one of
// V (dP^T = V dO^T): every fragment read feeds the MFMAs of both 16-query blocks of the wave.
template <int HDPV, bool DROP, int NW, int NST, int WPE = 2>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn_bwd_dq_wide_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NQ = 2, NKS = HDPV / 32, NNB = HDPV / 16;
  constexpr int IMG = 64 * HDPV * 2, PAIR = 2 * IMG;
  constexpr int QPB = NW * 16 * NQ;
  constexpr int DPT = 2 * (HDPV / 64) * (64 / 8 / NW);  // LDS-DMA instructions per wave and tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  int tile, h, b;
  if (!attn_tile_of_block((a.Tq + QPB - 1) / QPB, a.H, a.B, tile, h, b)) return;
  const int hd = a.hd;
  const unsigned short* Q = a.Q + b * a.sqb + h * hd;
  const unsigned short* K = a.K + b * a.skb + h * hd;
  const unsigned short* V = a.V + b * a.svb + h * hd;
  const unsigned short* dO = a.dO + b * a.sdob + h * hd;
  const int q0 = tile * QPB + wave * 16 * NQ;
  int qi[NQ], qrow[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    qi[j] = q0 + 16 * j + r;
    qrow[j] = qi[j] < a.Tq ? qi[j] : a.Tq - 1;
  }
  int kl = a.Tk;
  if (a.klen) kl = a.klen[b] < kl ? a.klen[b] : kl;
  int ntile = (kl + 63) / 64;
  if (a.causal) {
    const int last = (tile * QPB + QPB - 1) / 64 + 1;  // keys beyond the workgroup's last query are masked
    ntile = ntile < last ? ntile : last;
  }
  KImgFast<64, HDPV, NW> kfast, vfast;
  kfast.init(a.ldk, wave, lane);
  vfast.init(a.ldv, wave, lane);
  const int nfast = fast_tiles(a.Tk, 64);
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* img = smem + stage * PAIR;
    if (kt < nfast) {
      kfast.issue(img, K + (int64_t)kt * 64 * a.ldk, wave);
      vfast.issue(img + IMG, V + (int64_t)kt * 64 * a.ldv, wave);
    } else {
      load_kmajor_image<64, HDPV, NW>(img, K, a.ldk, kt * 64, a.Tk, hd, wave, lane);
      load_kmajor_image<64, HDPV, NW>(img + IMG, V, a.ldv, kt * 64, a.Tk, hd, wave, lane);
    }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < ntile) issue(t, t);
  // all HDPV/32 k-steps and HDPV/16 column blocks are computed: dims >= hd are zero in every LDS image
  bf16x8_t qf[NQ][NKS], dof[NQ][NKS];
  float lse2[NQ], dq_row[NQ];
  const unsigned short* Oh = a.O + b * a.sob + h * hd;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    // D = rowsum(dO * O) of the lane's query is taken here, from the dO fragments the kernel holds anyway and the
    // matching O fragments (each lane group has a quarter of the row), and written for the dK/dV kernel, which runs
    // after this one: no separate pass over dO and O (attn_bwd_prep_kernel serves the 64-query path only)
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[j][ks] = load_rowfrag(Q, a.ldq, qrow[j], ks, lane, hd);
      dof[j][ks] = load_rowfrag(dO, a.lddo, qrow[j], ks, lane, hd);
      const bf16x8_t of = load_rowfrag(Oh, a.ldo, qrow[j], ks, lane, hd);
      const u16x8_t x = __builtin_bit_cast(u16x8_t, dof[j][ks]), y = __builtin_bit_cast(u16x8_t, of);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum = fmaf(bf2f(x[e]), bf2f(y[e]), dsum);
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    dq_row[j] = dsum;
    if (g == 0 && qi[j] < a.Tq) const_cast<float*>(a.Dq)[((int64_t)b * a.H + h) * a.Tqp + qi[j]] = dsum;
    lse2[j] = a.lse[((int64_t)b * a.H + h) * a.Tqp + qrow[j]] * LOG2E;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (these loads are younger than the first tiles)
  f32x4_t acc[NQ][NNB];
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) acc[j][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float scale = a.scale;
  const float c2 = scale * LOG2E;
  auto tile_body = [&](auto full_c, int kt, int stage) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    const char* Kk = smem + stage * PAIR;
    const char* Vk = Kk + IMG;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float ds[NQ][8];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        f32x4_t sacc[NQ], pacc[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) sacc[j] = pacc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int row = 32 * s + rowperm(bb, r);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8_t kfr = kimg_frag<64>(Kk, row, ks, lane), vfr = kimg_frag<64>(Vk, row, ks, lane);
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            sacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[j][ks], sacc[j], 0, 0, 0);
            pacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, dof[j][ks], pacc[j], 0, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          unsigned keep = 0xFu;
          if constexpr (DROP)
            keep = ca_dropout_keep4(a.drop_seed, attn_drop_index(a, b, h, qrow[j], kt * 64 + 32 * s + 8 * g + 4 * bb), a.drop_p);
          const float kscale = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dpv = DROP ? (((keep >> e) & 1u) ? pacc[j][e] * kscale : 0.f) : pacc[j][e];
            float dsv = __builtin_amdgcn_exp2f(fmaf(sacc[j][e], c2, -lse2[j])) * (dpv - dq_row[j]);  // (x scale: when dQ is stored)
            if constexpr (!FULL) {
              const int key = kt * 64 + 32 * s + 8 * g + 4 * bb + e;
              const bool ok = key < kl && (!a.causal || key <= qi[j]);
              dsv = ok ? dsv : 0.f;
            }
            ds[j][4 * bb + e] = dsv;
          }
        }
      }
      bf16x8_t dsf[NQ];
#pragma unroll
      for (int j = 0; j < NQ; ++j) dsf[j] = pack8(ds[j]);
      // transposed key fragments in batches of four column blocks, the next batch in flight while one is used
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = timg_frag_k_async<64>(Kk, s, i, lane);
#pragma unroll
      for (int bt = 0; bt < NNB / 4; ++bt) {
        lds_wait_all();
#pragma unroll
        for (int i = 0; i < 4; ++i) tie((bt & 1) ? fb[i] : fa[i]);
        if (bt + 1 < NNB / 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ((bt & 1) ? fa[i] : fb[i]) = timg_frag_k_async<64>(Kk, s, 4 * (bt + 1) + i, lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            acc[j][4 * bt + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf[j], (bt & 1) ? fb[i] : fa[i], acc[j][4 * bt + i], 0, 0, 0);
      }
    }
  };
  int stage = 0;
  auto run = [&](auto full_c, int kt0, int kt1) __attribute__((always_inline)) {
    for (int kt = kt0; kt < kt1; ++kt) {
      const int ahead = ntile - 1 - kt;
      if (NST >= 4 && ahead >= 2)
        wait_vm<2 * DPT>();
      else if (NST >= 3 && ahead >= 1)
        wait_vm<DPT>();
      else
        wait_vm<0>();
      __syncthreads();
      if (kt + NST - 1 < ntile) issue(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);
      tile_body(full_c, kt, stage);
      stage = stage + 1 == NST ? 0 : stage + 1;
    }
  };
  const int nfull = a.causal ? 0 : (kl / 64 < ntile ? kl / 64 : ntile);
  run(std::true_type{}, 0, nfull);
  run(std::false_type{}, nfull, ntile);
  __syncthreads();  // every wave is done with the images: the LDS becomes output staging
  unsigned short* dQ = a.dQ + b * a.sdqb + h * hd;
  const float sc4[4] = {scale, scale, scale, scale};  // dS was formed without the softmax scale
  char* st = smem + wave * (16 * (HDPV * 2 + 16));
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (j) __builtin_amdgcn_wave_barrier();
    store_tile16<HDPV>(st, acc[j], sc4, dQ, a.lddq, q0 + 16 * j, a.Tq, hd, lane);
  }
}

// ---- C ABI ------------------------------------------------------------------------------------------------
static int attn_check(const CaAttnDesc* d, const char* who) {
  CA_CHECK_ARG(d && d->Q && d->K && d->V, "%s: null pointer", who);
  CA_CHECK_ARG(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "%s: bad shape", who);
  CA_CHECK_ARG(d->hd >= 8 && d->hd <= 128 && (d->hd % 8) == 0, "%s: head_dim must be a multiple of 8 in [8,128]", who);
  CA_CHECK_ARG((d->ldq % 8) == 0 && (d->ldk % 8) == 0 && (d->ldv % 8) == 0, "%s: strides must be multiples of 8", who);
  CA_CHECK_ARG(d->lse != nullptr && d->Tqp >= d->Tq && (d->Tqp % 32) == 0, "%s: lse needs Tqp %% 32 == 0 rows", who);
  CA_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "%s: bad dropout_p", who);
  return CA_OK;
}
static AttnArgs to_args(const CaAttnDesc& d) {
  AttnArgs a;
  a.Q = (const unsigned short*)d.Q; a.K = (const unsigned short*)d.K; a.V = (const unsigned short*)d.V;
  a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.sqb = d.sqb; a.skb = d.skb; a.svb = d.svb;
  a.O = (unsigned short*)d.O; a.ldo = d.ldo; a.sob = d.sob;
  a.O8 = (unsigned char*)d.O8; a.o8_scale = d.o8_scale; a.o8_amax = (unsigned int*)d.o8_amax;
  a.lse = d.lse; a.klen = d.klen; a.B = d.B; a.H = d.H; a.Tq = d.Tq; a.Tk = d.Tk; a.hd = d.hd; a.Tqp = d.Tqp;
  a.causal = d.causal; a.scale = d.scale;
  a.dO = (const unsigned short*)d.dO; a.lddo = d.lddo; a.sdob = d.sdob; a.Dq = d.Dq;
  a.dQ = (unsigned short*)d.dQ; a.dK = (unsigned short*)d.dK; a.dV = (unsigned short*)d.dV;
  a.lddq = d.lddq; a.lddk = d.lddk; a.lddv = d.lddv; a.sdqb = d.sdqb; a.sdkb = d.sdkb; a.sdvb = d.sdvb;
  a.drop_p = d.dropout_p; a.drop_seed = d.dropout_seed;
  a.qp_x = nullptr; a.qp_ldx = 0; a.qp_gamma = a.qp_beta = a.qp_bias = nullptr; a.qp_W = nullptr; a.qp_ldw = 0; a.qp_d = 0;
  a.qp_eps = 0.f;
  a.nsplit = 1; a.split_slab = nullptr; a.split_cnt = nullptr;
  return a;
}

// Key split of the small-query kernel: on when the caller lent a workspace, the keys are many and (clips x heads) leaves
// at least half of the CUs without a workgroup.  Sets a.nsplit / the slab and counter pointers; returns the grid size.
// grids from which the single-query kernels run two workgroups per CU (rings of two tiles): more than 1.5 workgroups per
// CU (CA_SMALLQ_2PERCU: the grid size; 0 = never)
static unsigned smallq_two_per_cu() {
  static const unsigned thr = [] {
    const char* e = getenv("CA_SMALLQ_2PERCU");
    if (e) return (unsigned)(atoi(e) > 0 ? atoi(e) : 0x7fffffff);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return (unsigned)(n + n / 2);
  }();
  return thr;
}
// How many workgroups share one (clip, head)'s keys: CUs / (clips x heads), at most CA_ATTN_SPLIT_MAX (8 clips x 16 heads on
// 256 CUs: 2; 1 clip: 4; 16 x 16: 1).  ONE rule for ca_attn_fwd, ca_decode_attn_qproj and ca_whisper_decode_token
// (decode.hip): the merge order, and so the bits, are the same.  (Round 6, measured and dropped: splitting EVERY item where
// clips x heads exceeds the CUs so that the rounds come out even - whisper-large at 16 clips, 320 items -> 1280
// quarter-items - took the persistent token from 3.04 to 3.17 ms and the launch sequence from 4.5 to 6.3: every part pays
// the prologue, the barriers and the merge.)
// More items than CUs, fewer than two rounds of them (whisper-large at 16 clips: 320 on 256): the items of the second,
// partly filled round - and only they - are split so that their parts fill it: 64 items x 4 parts, a quarter of the keys
// each, beside 256 whole items.  Same box, per token: persistent launch 2.97 -> 2.74 ms, launch sequence 4.48 -> 4.02
// (NOTEBOOK R6.5); CA_ATTN_SPLIT_TAIL=0 keeps whole items.
CaKeySplit ca_attn_key_split(int bh, int ncu, int cap) {
  cap = cap > CA_ATTN_SPLIT_MAX ? CA_ATTN_SPLIT_MAX : cap;
  CaKeySplit s;
  s.ns = 1; s.tail_start = bh; s.ns_tail = 1;
  static const int tail_on = [] { const char* e = getenv("CA_ATTN_SPLIT_TAIL"); return e ? atoi(e) : 1; }();
  if (bh <= ncu) {
    int ns = ncu / bh;
    ns = ns > cap ? cap : ns;
    s.ns = ns < 2 ? 1 : ns;
  } else if (bh < 2 * ncu && tail_on) {
    const int t = bh - ncu;
    int nt = ncu / t;
    nt = nt > cap ? cap : nt;
    if (nt >= 2) {
      s.tail_start = ncu;
      s.ns_tail = nt;
    }
  }
  return s;
}
static int smallq_split(const CaAttnDesc& d, AttnArgs& a, unsigned& grid) {
  const int bh = d.B * d.H;
  grid = (unsigned)bh;
  a.nsplit = 1;
  if (!d.split_ws || d.Tk < 1024) return CA_OK;
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  static const int cap = [] { const char* e = getenv("CA_ATTN_SPLIT"); return e ? atoi(e) : CA_ATTN_SPLIT_MAX; }();
  const CaKeySplit sp = ca_attn_key_split(bh, ncu, cap);
  if (sp.ns < 2 && sp.ns_tail < 2) return CA_OK;
  CA_CHECK_ARG(d.split_ws_bytes >= CA_ATTN_SPLIT_WS_BYTES(d.B, d.H) && ((uintptr_t)d.split_ws % 16) == 0,
               "attention: split_ws needs CA_ATTN_SPLIT_WS_BYTES(B, H) bytes, 16-byte aligned");
  a.nsplit = sp.ns > sp.ns_tail ? sp.ns : sp.ns_tail;
  a.split = sp;
  a.split_cnt = (unsigned int*)d.split_ws;
  a.split_slab = (float*)((char*)d.split_ws + ((size_t)bh * 4 + 255) / 256 * 256);
  grid = (unsigned)ca_key_split_parts(sp, bh);
  return CA_OK;
}

extern "C" int ca_attn_fwd(const CaAttnDesc* desc, void* stream) {
  if (int rc = attn_check(desc, "ca_attn_fwd")) return rc;
  CA_CHECK_ARG(desc->O != nullptr, "ca_attn_fwd: null output");
  static const int wide_on = [] { const char* e = getenv("CA_ATTN_WIDE"); return e ? atoi(e) : 1; }();
  CA_CHECK_ARG(desc->O8 == nullptr || (desc->o8_scale != nullptr && desc->Tq >= 100 && wide_on && (desc->ldo % 8) == 0 &&
                                       ((uintptr_t)desc->O8 % 8) == 0),
               "ca_attn_fwd: O8 needs a scale, >= 100 queries per head (the wide kernels) and 8-byte aligned rows");
  const AttnArgs a = to_args(*desc);
  dim3 grid(attn_grid((desc->Tq + 63) / 64, desc->H, desc->B)), block(256);
  hipStream_t s = (hipStream_t)stream;
  // 128-query workgroups when the query side fills them; CA_ATTN_WIDE=0 keeps the 64-query kernel
  static const int wide = [] { const char* e = getenv("CA_ATTN_WIDE"); return e ? atoi(e) : 1; }();
  if (wide && desc->Tq >= 100) {
    const bool drop = desc->dropout_p > 0.f;
#define CA_FWD_WIDE(HDPV, DROP)                                                                                   \
  hipLaunchKernelGGL((attn_fwd_wide_kernel<HDPV, DROP, 4, 2>), dim3(attn_grid((desc->Tq + 127) / 128, desc->H, desc->B)), \
                     dim3(256), 2 * 2 * 64 * HDPV * 2, s, a)
    // head_dim <= 64 without dropout: three workgroups per CU (see the kernel's header); CA_ATTN_WPE3=0 keeps two (A/B)
    static const int wpe3 = [] { const char* e = getenv("CA_ATTN_WPE3"); return e ? atoi(e) : 1; }();
    if (desc->hd <= 64 && wpe3 && !drop) {
      hipLaunchKernelGGL((attn_fwd_wide_kernel<64, false, 4, 2, 3>), dim3(attn_grid((desc->Tq + 127) / 128, desc->H, desc->B)),
                         dim3(256), 2 * 2 * 64 * 64 * 2, s, a);
      CA_CHECK_LAUNCH("ca_attn_fwd");
      return CA_OK;
    }
    if (desc->hd <= 64) {
      if (drop) CA_FWD_WIDE(64, true); else CA_FWD_WIDE(64, false);
    } else {
      if (drop) CA_FWD_WIDE(128, true); else CA_FWD_WIDE(128, false);
    }
#undef CA_FWD_WIDE
    CA_CHECK_LAUNCH("ca_attn_fwd");
    return CA_OK;
  }
  if (desc->dropout_p > 0.f) {  // training with dropout on the attention probabilities: the general kernels only
    if (desc->hd <= 64)
      hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, block, 2 * 64 * 64 * 2, s, a);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, block, 2 * 64 * 128 * 2, s, a);
    CA_CHECK_LAUNCH("ca_attn_fwd");
    return CA_OK;
  }
  if (desc->Tq <= 16 && desc->hd <= 64 && !desc->causal) {  // greedy decoding: the waves split the keys
    constexpr int SMALLQ_LDS = 4 * 3 * 64 * 64 * 2;  // four waves x ring of three 8-KiB V images
    constexpr int SMALLQ_LDS2 = 4 * 2 * 64 * 64 * 2;  // ... of two: two workgroups per CU (large batches)
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute((const void*)attn_fwd_smallq_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, SMALLQ_LDS);
      hipFuncSetAttribute((const void*)attn_fwd_smallq_kernel<64, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMALLQ_LDS2);
      attr = true;
    }
    AttnArgs as = a;
    unsigned sgrid;
    if (int rc = smallq_split(*desc, as, sgrid)) return rc;
    if (sgrid >= smallq_two_per_cu())
      hipLaunchKernelGGL((attn_fwd_smallq_kernel<64, false, 2>), dim3(sgrid), block, SMALLQ_LDS2, s, as);
    else
      hipLaunchKernelGGL((attn_fwd_smallq_kernel<64>), dim3(sgrid), block, SMALLQ_LDS, s, as);
    CA_CHECK_LAUNCH("ca_attn_fwd");
    return CA_OK;
  }
  if (desc->hd <= 64)
    hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, block, 2 * 64 * 64 * 2, s, a);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<128>), grid, block, 2 * 64 * 128 * 2, s, a);
  CA_CHECK_LAUNCH("ca_attn_fwd");
  return CA_OK;
}

// Greedy decoding: LayerNorm + query projection + single-query attention over a K|V cache in one launch per layer.
extern "C" int ca_decode_attn_qproj(const CaAttnDesc* desc, const void* x, int64_t ldx, const float* ln_gamma,
                                    const float* ln_beta, float ln_eps, const void* Wq, int64_t ldw, const float* bq,
                                    int32_t d_model, void* stream) {
  CA_CHECK_ARG(desc && x && ln_gamma && ln_beta && Wq && bq, "ca_decode_attn_qproj: null pointer");
  CA_CHECK_ARG(desc->K && desc->V && desc->O && desc->B > 0 && desc->H > 0 && desc->Tk > 0, "ca_decode_attn_qproj: bad descriptor");
  CA_CHECK_ARG(desc->Tq == 1 && desc->hd <= 64 && (desc->hd % 8) == 0 && !desc->causal && desc->dropout_p == 0.f,
               "ca_decode_attn_qproj: one query per clip, head_dim <= 64, no mask, no dropout");
  CA_CHECK_ARG(d_model >= 64 && d_model <= 2048 && (d_model % 8) == 0 && (ldx % 8) == 0 && (ldw % 8) == 0 &&
                   ((uintptr_t)x % 16) == 0 && ((uintptr_t)Wq % 16) == 0 && ((uintptr_t)ln_gamma % 16) == 0 &&
                   ((uintptr_t)ln_beta % 16) == 0,
               "ca_decode_attn_qproj: d_model in [64, 2048], multiples of 8, 16-byte aligned operands");
  CA_CHECK_ARG((desc->ldk % 8) == 0 && (desc->ldv % 8) == 0, "ca_decode_attn_qproj: ldk / ldv must be multiples of 8");
  AttnArgs a = to_args(*desc);
  a.qp_x = (const unsigned short*)x;
  a.qp_ldx = ldx;
  a.qp_gamma = ln_gamma;
  a.qp_beta = ln_beta;
  a.qp_eps = ln_eps;
  a.qp_W = (const unsigned short*)Wq;
  a.qp_ldw = ldw;
  a.qp_bias = bq;
  a.qp_d = d_model;
  constexpr int LDS = 4 * 3 * 64 * 64 * 2 + 4096 + 1024;  // the V rings + the normalised row + the partial products
  constexpr int LDS2 = 4 * 2 * 64 * 64 * 2 + 4096 + 1024;  // rings of two: two workgroups per CU (large batches)
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_fwd_smallq_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)attn_fwd_smallq_kernel<64, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
    attr = true;
  }
  unsigned sgrid;
  if (int rc = smallq_split(*desc, a, sgrid)) return rc;
  // (non-temporal loads of the K|V stream measured slower here: 8 491 -> 8 307 audio-s/s at 128 clips, round 5)
  if (sgrid >= smallq_two_per_cu())
    hipLaunchKernelGGL((attn_fwd_smallq_kernel<64, true, 2>), dim3(sgrid), dim3(256), LDS2, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((attn_fwd_smallq_kernel<64, true>), dim3(sgrid), dim3(256), LDS, (hipStream_t)stream, a);
  CA_CHECK_LAUNCH("ca_decode_attn_qproj");
  return CA_OK;
}

extern "C" int ca_attn_bwd(const CaAttnDesc* desc, void* stream) {
  if (int rc = attn_check(desc, "ca_attn_bwd")) return rc;
  CA_CHECK_ARG(desc->dO && desc->O && desc->Dq && desc->dQ && desc->dK && desc->dV, "ca_attn_bwd: null pointer");
  CA_CHECK_ARG((desc->lddo % 8) == 0, "ca_attn_bwd: lddo must be a multiple of 8");
  const AttnArgs a = to_args(*desc);
  hipStream_t s = (hipStream_t)stream;
  static const int wide = [] { const char* e = getenv("CA_ATTN_WIDE"); return e ? atoi(e) : 1; }();
  const bool drop = desc->dropout_p > 0.f;
  const bool dq_wide = wide && desc->Tq >= 100;
  dim3 gk(attn_grid((desc->Tk + 63) / 64, desc->H, desc->B)), gq(attn_grid((desc->Tq + 63) / 64, desc->H, desc->B)), block(256);
  // dQ.  128-query workgroups when the query side fills them: that kernel also produces D = rowsum(dO * O) for the
  // dK/dV kernel behind it.  Otherwise D comes from its own pass first.
  if (!dq_wide) {
    const int64_t groups = (int64_t)desc->B * desc->H * desc->Tq;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0, s,
                       (const unsigned short*)desc->dO, desc->lddo, desc->sdob, (const unsigned short*)desc->O,
                       desc->ldo, desc->sob, (float*)desc->Dq, desc->H, desc->Tq, desc->Tqp, desc->hd, desc->B);
  }
  static const int dq_wpe3 = [] { const char* e = getenv("CA_ATTN_DQ_WPE3"); return e ? atoi(e) : 1; }();
  if (dq_wide && dq_wpe3 && desc->hd <= 64 && !drop) {
    hipLaunchKernelGGL((attn_bwd_dq_wide_kernel<64, false, 4, 2, 3>), dim3(attn_grid((desc->Tq + 127) / 128, desc->H, desc->B)),
                       dim3(256), 2 * 2 * 64 * 64 * 2, s, a);
  } else
#define CA_DQ(HDPV, DROP)                                                                                          \
  do {                                                                                                             \
    if (dq_wide)                                                                                                   \
      hipLaunchKernelGGL((attn_bwd_dq_wide_kernel<HDPV, DROP, 4, 2>),                                              \
                         dim3(attn_grid((desc->Tq + 127) / 128, desc->H, desc->B)), dim3(256), 2 * 2 * 64 * HDPV * 2, s, a); \
    else                                                                                                           \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<HDPV, DROP>), gq, block, 2 * 64 * HDPV * 2, s, a);                    \
  } while (0)
  if (desc->hd <= 64) {
    if (drop) CA_DQ(64, true); else CA_DQ(64, false);
  } else {
    if (drop) CA_DQ(128, true); else CA_DQ(128, false);
  }
#undef CA_DQ
  // dK, dV: 64 keys per workgroup (a 128-key, 8-wave variant with a 4-deep ring measured no faster at T = 499 and 5 %
  // slower at T = 1500: dropped)
  // ... and a wave that owns two key blocks (128 keys per workgroup) halves the LDS fragment reads per MFMA: see
  // attn_bwd_dkv_wide_kernel.  CA_ATTN_DKV_WIDE=0 keeps the 64-key kernel.
  static const int dkv_wide_on = [] { const char* e = getenv("CA_ATTN_DKV_WIDE"); return e ? atoi(e) : 1; }();
  // Measured (tools/archive/exp_dkv.sh, rocprofv3 per-kernel averages): head_dim <= 64 (XLS-R-300M, Whisper) 226 -> 208 us over
  // the models' shapes (XLS-R-300M backward 72.5 -> 63.1 us per layer, whisper-large-turbo encoder 523 -> 494); at
  // head_dim 80 / 120 the doubled accumulators (374 registers) leave one wave per SIMD and it LOSES (73.3 -> 83.7 us):
  // the 64-key kernel stays there.
  const bool dkv_wide = dkv_wide_on && desc->Tk >= 100 && desc->hd <= 64;
  static const int dkv_wpe3 = [] { const char* e = getenv("CA_ATTN_DKV_WPE3"); return e ? atoi(e) : 0; }();
  if (dkv_wide && dkv_wpe3 && desc->hd <= 64 && !drop) {
    hipLaunchKernelGGL((attn_bwd_dkv_wide_kernel<64, false, 2, 3>), dim3(attn_grid((desc->Tk + 127) / 128, desc->H, desc->B)),
                       block, 4 * 32 * 64 * 2, s, a);
  } else
#define CA_DKV(HDPV, DROP)                                                                                            \
  do {                                                                                                               \
    if (dkv_wide)                                                                                                    \
      hipLaunchKernelGGL((attn_bwd_dkv_wide_kernel<HDPV, DROP, 2>),                                                  \
                         dim3(attn_grid((desc->Tk + 127) / 128, desc->H, desc->B)), block, 4 * 32 * HDPV * 2, s, a); \
    else                                                                                                             \
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDPV, DROP>), gk, block, 4 * 32 * HDPV * 2, s, a);                     \
  } while (0)
  if (desc->hd <= 64) {
    if (drop) CA_DKV(64, true); else CA_DKV(64, false);
  } else {
    if (drop) CA_DKV(128, true); else CA_DKV(128, false);
  }
#undef CA_DKV
  CA_CHECK_LAUNCH("ca_attn_bwd");
  return CA_OK;
}
