// MFMA bf16 GEMM family for gfx950 (CDNA4).  See include/coral_amd.h (ca_gemm_bf16).
//
// Design (v1, "128x128x64 / LDS-DMA / 2-buffer"):
//   * 256 threads = 4 waves in a 2x2 grid, each wave owns a 64x64 output tile as 4x4
//     v_mfma_f32_16x16x32_bf16 accumulators (operands swapped so a lane ends up with 4
//     consecutive n of one m: 8-byte bf16 / 16-byte fp32 stores).
//   * both operand tiles are staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round
//     trip); the LDS image is lane-linear, bank conflicts are removed by permuting the
//     per-lane SOURCE chunk and applying the same XOR on the fragment read.
//   * KMAJOR operands ([rows][64 k], 128-B rows) are read with ds_read_b128;
//     MNMAJOR operands ([64 k][128 mn], 256-B rows) with ds_read_b64_tr_b16 (hardware
//     transpose), so NT / NN / TN / TT all run natively without transposed copies.
//   * out-of-range rows are clamped (results discarded), out-of-range k reads a zero page.
#include "common.h"
#include <type_traits>
#include <cstdlib>

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand tile
#define STAGE_BYTES (2 * TILE_BYTES)
#define NSTAGE 2
#define EPI_PITCH 68  // floats; epilogue staging row pitch (272 B)
#define LDS_BYTES (4 * 64 * EPI_PITCH * 4)  // 69632 >= NSTAGE*STAGE_BYTES

__device__ __attribute__((aligned(16))) uint32_t g_ca_zero_page[4];
__device__ int g_ca_epi_general = 0;  // tests: 1 = every wave tile takes the general epilogue walk (ca_gemm_debug_general_epilogue)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA, 16 B per lane from a per-lane address to the wave's 1-KiB LDS piece.  Inline asm like the scalar-base
// form below, so that M0 is only ever written inside these statements (the compiler does not track M0 across them).
// The hazard recogniser does not look inside an asm string, so the wait state between the SALU write of M0 and the
// LDS-DMA that reads it is written out (hipcc pads its own LDS-DMA the same way).  The scalar-base form's other
// hazard - a VMEM instruction reading SGPRs that a VALU instruction (v_readfirstlane) wrote needs five states - does
// not arise here: every base below is the result of scalar arithmetic on kernel arguments and tile indices.
__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(g), "s"((uint32_t)(uintptr_t)(lptr_t)lds_wave_base)
               : "memory", "m0");
}

// ---- per-lane loader state -------------------------------------------------------------
// NI = LDS-DMA instructions per wave per tile (each moves 1 KiB = 64 lanes x 16 B).
template <int NI>
struct KMajorLoader {  // operand stored [row][k], k contiguous; LDS image [rows][64 k], 128-B rows
  const char* p[NI];   // current source (advanced BK elements per step)
  int kc[NI];          // element offset of this lane's chunk inside the k-step
  __device__ __forceinline__ void init(const __bf16* base, int64_t ld, int row0, int nrows,
                                       int wave, int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = (wave * NI + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int rr = row0 + r;
      rr = rr < nrows ? rr : nrows - 1;
      kc[i] = c * 8;
      p[i] = (const char*)(base + (int64_t)rr * ld + c * 8);
    }
  }
  __device__ __forceinline__ void issue(char* tile, int wave, int k0, int K) {
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_one(tile, wave, k0, K, i);
  }
  __device__ __forceinline__ void issue_one(char* tile, int wave, int k0, int K, int i) {
    const void* src = (k0 + kc[i] < K) ? (const void*)p[i] : (const void*)g_ca_zero_page;
    glds16(src, tile + (wave * NI + i) * 1024);
    p[i] += BK * 2;
  }
};

// operand stored [k][mn], mn contiguous; LDS image [64 k][PC*8 mn]; PC = 16-B chunks per row.
// The source pointer of each lane walks down the k rows by plain 64-bit adds (BK rows per step);
// with segmented rows (kseg > 0) it hops by (segstride - kseg*ld) whenever it crosses a segment.
template <int NI, int PC, bool KS = true>
struct MNMajorLoader {
  const char* p[NI];  // current source address of this lane's chunk
  int t[NI];          // row index inside the current segment
  int64_t step, hop;  // bytes per BK rows; extra bytes when crossing a segment boundary
  int kseg;
  static constexpr int RPI = 64 / PC;  // k-rows per LDS-DMA instruction
  __device__ __forceinline__ void init(const __bf16* base, int64_t ld, int kseg_,
                                       int64_t segstride, int col0, int ncols, int wave,
                                       int lane) {
    kseg = kseg_;
    step = (int64_t)BK * ld * 2;
    hop = kseg > 0 ? (segstride - (int64_t)kseg * ld) * 2 : 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kr = (wave * NI + i) * RPI + lane / PC;
      const int swz = (kr & 3) | (((kr >> 3) & 1) << 2);
      const int c = (lane % PC) ^ (swz << 1);
      int cc = col0 + c * 8;
      const int nc8 = (ncols + 7) & ~7;  // rows are readable up to ncols rounded up to 8
      cc = cc <= nc8 - 8 ? cc : nc8 - 8;
      int seg = 0, tt = kr;
      if (kseg > 0) {
        seg = kr / kseg;
        tt = kr % kseg;
      }
      t[i] = tt;
      p[i] = (const char*)(base + cc) + ((int64_t)seg * segstride + (int64_t)tt * ld) * 2;
    }
  }
  __device__ __forceinline__ void issue(char* tile, int wave, int lane, int k0, int K) {
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_one(tile, wave, lane, k0, K, i);
  }
  __device__ __forceinline__ void issue_one(char* tile, int wave, int lane, int k0, int K, int i) {
    const int kr = (wave * NI + i) * RPI + lane / PC;
    const void* src = (k0 + kr < K) ? (const void*)p[i] : (const void*)g_ca_zero_page;
    glds16(src, tile + (wave * NI + i) * 1024);
    p[i] += step;
    if (KS && kseg > 0) {
      t[i] += BK;
      while (t[i] >= kseg) {
        t[i] -= kseg;
        p[i] += hop;
      }
    }
  }
};

// LDS-DMA with the scalar-base addressing form: 16 B per lane from base + off (off: 32-bit, zero-extended) to the
// wave's 1-KiB LDS piece at byte address lds_piece (wave-uniform, goes through M0)
__device__ __forceinline__ void glds16_sbase(const char* base, uint32_t off, uint32_t lds_piece) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(off), "s"(base), "s"(lds_piece)
               : "memory", "m0");
}
// ---- loaders of kernel X: one wave-uniform base per operand tile + a 32-bit offset per lane and piece ----------
// The base walks down K by scalar adds; a full K-step issues its LDS-DMA without any per-lane address arithmetic or
// predicate.  Only the last, partial K-step (K % 64 != 0) selects the zero page per lane.  Rows / columns beyond
// the operand are clamped to its last one (their products are never stored).
template <int NI>
struct KMajorStream {  // operand stored [row][k], k contiguous; LDS image [rows][64 k], 128-B rows
  const char* base;    // wave-uniform: first row of the tile, current k
  uint32_t off[NI];    // byte offset of this lane's 16-B chunk
  int kc[NI];          // element offset of the chunk inside the K-step (tail predicate)
  __device__ __forceinline__ void init(const __bf16* b, int64_t ld, int row0, int nrows, int wave, int lane) {
    base = (const char*)(b + (int64_t)row0 * ld);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = (wave * NI + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int rr = row0 + r;
      rr = (rr < nrows ? rr : nrows - 1) - row0;
      kc[i] = c * 8;
      off[i] = (uint32_t)(((int64_t)rr * ld + c * 8) * 2);
    }
  }
  // one-byte elements (fp8): the same 128-B rows hold 128 k
  __device__ __forceinline__ void init_bytes(const char* b, int64_t ld, int row0, int nrows, int wave, int lane) {
    base = b + (int64_t)row0 * ld;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = (wave * NI + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int rr = row0 + r;
      rr = (rr < nrows ? rr : nrows - 1) - row0;
      kc[i] = c * 16;
      off[i] = (uint32_t)((int64_t)rr * ld + c * 16);
    }
  }
  __device__ __forceinline__ void issue_one(char* tile, int wave, int i) {
    glds16_sbase(base, off[i], (uint32_t)(uintptr_t)(lptr_t)(tile + (wave * NI + i) * 1024));
  }
  // partial K-step: chunks beyond K load some valid bytes instead (offset 0) and are zeroed in LDS by zero_fix()
  __device__ __forceinline__ void issue_one_tail(char* tile, int wave, int krem, int i) {
    glds16_sbase(base, kc[i] < krem ? off[i] : 0u, (uint32_t)(uintptr_t)(lptr_t)(tile + (wave * NI + i) * 1024));
  }
  // after this wave's LDS-DMA of the partial K-step has landed, before the barrier that publishes the tile
  __device__ __forceinline__ void zero_fix(char* tile, int wave, int lane, int krem) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (kc[i] >= krem) *(uint4*)(tile + (wave * NI + i) * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
  }
  __device__ __forceinline__ void advance() { base += BK * 2; }
};
template <int NI, int PC>
struct MNMajorStream {  // operand stored [k][mn], mn contiguous (plain rows, no segments); LDS image [64 k][PC*8 mn]
  const char* base;     // wave-uniform: column col0 of row k0
  uint32_t off[NI];
  int64_t step;
  static constexpr int RPI = 64 / PC;  // k-rows per LDS-DMA instruction
  __device__ __forceinline__ void init(const __bf16* b, int64_t ld, int col0, int ncols, int wave, int lane) {
    base = (const char*)(b + col0);
    step = (int64_t)BK * ld * 2;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kr = (wave * NI + i) * RPI + lane / PC;
      const int swz = (kr & 3) | (((kr >> 3) & 1) << 2);
      const int c = (lane % PC) ^ (swz << 1);
      int cc = col0 + c * 8;
      const int nc8 = (ncols + 7) & ~7;  // rows are readable up to ncols rounded up to 8
      cc = (cc <= nc8 - 8 ? cc : nc8 - 8) - col0;
      off[i] = (uint32_t)(((int64_t)kr * ld + cc) * 2);
    }
  }
  __device__ __forceinline__ void issue_one(char* tile, int wave, int i) {
    glds16_sbase(base, off[i], (uint32_t)(uintptr_t)(lptr_t)(tile + (wave * NI + i) * 1024));
  }
  __device__ __forceinline__ void issue_one_tail(char* tile, int wave, int lane, int krem, int i) {
    const int kr = (wave * NI + i) * RPI + lane / PC;
    glds16_sbase(base, kr < krem ? off[i] : 0u, (uint32_t)(uintptr_t)(lptr_t)(tile + (wave * NI + i) * 1024));
  }
  __device__ __forceinline__ void zero_fix(char* tile, int wave, int lane, int krem) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kr = (wave * NI + i) * RPI + lane / PC;
      if (kr >= krem) *(uint4*)(tile + (wave * NI + i) * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void advance() { base += step; }
};

// ---- fragment reads ----------------------------------------------------------------------
// KMAJOR tile: 16 rows starting at rb, k-step s (32 k): lane gets row rb+(lane&15),
// k = 32 s + 8 (lane>>4) .. +7.
__device__ __forceinline__ bf16x8_t frag_kmajor(const char* tile, int rb, int s, int lane) {
  const int r = rb + (lane & 15);
  const int c = (4 * s + (lane >> 4)) ^ ((r >> 1) & 7);
  return *(const bf16x8_t*)(tile + r * 128 + c * 16);
}
// same fragment, issued as inline asm (completion through lds_wait): lets a kernel order its reads
// against its MFMA blocks by hand
__device__ __forceinline__ bf16x8_t frag_kmajor_async(const char* tile, int rb, int s, int lane) {
  const int r = rb + (lane & 15);
  const int c = (4 * s + (lane >> 4)) ^ ((r >> 1) & 7);
  const uint32_t a = (uint32_t)(uintptr_t)(lptr_t)(tile + r * 128 + c * 16);
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}
// MNMAJOR tile (row pitch PITCH bytes): 16 columns starting at cb (multiple of 16), k-step s:
// two transposed reads (k = 8g..8g+3 and 8g+4..8g+7 of the 32-k step).
// The reads are inline asm on purpose: hipcc treats the ds_read_tr builtin as a memory operation
// that may alias the LDS-DMA of the NEXT tile and puts `s_waitcnt vmcnt(0)` in front of it, which
// serialises the whole prefetch behind every fragment read (seen in the ISA of every MN-major
// variant).  The price: the compiler does not know when the data arrives, so every consumer must
// pass the fragments through lds_wait() first.
template <int PITCH>
__device__ __forceinline__ bf16x8_t frag_mnmajor(const char* tile, int cb, int s, int lane) {
  const int g = lane >> 4;
  const int q = (lane & 15) >> 2;
  const int p = lane & 3;
  const int kr = 32 * s + 8 * g + q;
  const int swz = q | ((g & 1) << 2);
  const int c = ((cb >> 3) + (p >> 1)) ^ (swz << 1);
  const uint32_t a0 = (uint32_t)(uintptr_t)(lptr_t)(tile + kr * PITCH + c * 16 + (p & 1) * 8);
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a0), "n"(4 * PITCH));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// fragment reads with the stage / k-half / fragment folded into the immediate offset (completion through lds_wait)
template <int OFF>
__device__ __forceinline__ bf16x8_t lds_read_b128(uint32_t a) {
  static_assert(OFF >= 0 && OFF < 65536, "DS immediate offset");
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <int OFF, int HI>
__device__ __forceinline__ bf16x8_t lds_read_tr(uint32_t a) {
  static_assert(OFF >= 0 && OFF + HI < 65536, "DS immediate offset");
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + HI));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// all outstanding LDS reads have returned; the "+v" ties make every later use of f wait for it
__device__ __forceinline__ void lds_wait(bf16x8_t (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
}

// XCD-aware tile rasterisation.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and
// b+8 share an L2), and the tiles that are resident together on one XCD should share operand
// panels: they stream the same K-slices at about the same time, so each slice is pulled across
// the fabric once per XCD and served to the other tiles from that XCD's 4-MiB L2.  The grid is
// cut into super-blocks of SBM x SBN tiles (= the number of tiles one XCD holds at once); the
// i-th workgroup of XCD x works on tile (i % (SBM*SBN)) of super-block (i / (SBM*SBN))*8 + x.
// Placement only affects speed: any dispatch order gives the same result.
// Small / medium grids: the tile grid is cut into exactly 8 rectangular blocks, one per XCD (bm x bn blocks with
// bm * bn = 8, the split that minimises block height + width = the operand bands an XCD has to stream).
// xcd_split returns bm; the block is hb x wb tiles.
__host__ __device__ __forceinline__ int xcd_split(int ntm, int ntn, int& hb, int& wb) {
  int best = 1, cost = 1 << 30;
#pragma unroll
  for (int bm = 1; bm <= 8; bm *= 2) {
    const int bn = 8 / bm;
    const int c = (ntm + bm - 1) / bm + (ntn + bn - 1) / bn;
    if (c < cost) {
      cost = c;
      best = bm;
    }
  }
  hb = (ntm + best - 1) / best;
  wb = (ntn + 8 / best - 1) / (8 / best);
  return best;
}
// Tile `l` of a grid in run order: the shorter grid dimension runs fastest, so a run of consecutive tiles covers a
// compact band of the grid.
__device__ __forceinline__ void run_tile(int l, int ntm, int ntn, int& tm, int& tn) {
  if (ntn <= ntm) {
    tm = l / ntn;
    tn = l % ntn;
  } else {
    tn = l / ntm;
    tm = l % ntm;
  }
}
// The rectangular split leaves XCDs unevenly loaded when the grid does not divide (5 x 5 tiles: 6, 6, 3, 0, 4, 4, 2,
// 0 per XCD): where it would pad by more than a quarter, each XCD takes a run of ceil(T / 8) consecutive tiles.
// bal (CaGemmDesc.xcd_balanced, set by the library beside a resident collective): always runs - every XCD gets
// ceil(T / 8) tiles.  The rectangular split may hand one XCD exactly its 32 CUs' worth of a 240-tile grid (32, 32, ...,
// 24, 24): with two CUs of an XCD held by another kernel that XCD runs a second round and the launch takes twice as long.
__host__ __device__ __forceinline__ bool xcd_use_runs(int ntm, int ntn, int hb, int wb, bool bal = false) {
  return bal || 8 * hb * wb * 4 > ntm * ntn * 5;
}
__host__ __device__ __forceinline__ int xcd_grid(int ntm, int ntn, bool bal = false) {
  if (ntm * ntn <= 8) return ntm * ntn;  // a handful of tiles (batched attention-sized problems): plain numbering
  int hb, wb;
  xcd_split(ntm, ntn, hb, wb);
  if (xcd_use_runs(ntm, ntn, hb, wb, bal)) return 8 * ((ntm * ntn + 7) / 8);
  return 8 * hb * wb;
}
// tile of block `bid` under that split (false = padding block)
__device__ __forceinline__ bool xcd_tile(int bid, int ntm, int ntn, int& tm, int& tn, bool bal = false) {
  if (ntm * ntn <= 8) {
    tm = bid / ntn;
    tn = bid % ntn;
    return true;
  }
  int hb, wb;
  const int bm = xcd_split(ntm, ntn, hb, wb);
  const int bn = 8 / bm;
  const int x = bid & 7, idx = bid >> 3;
  if (xcd_use_runs(ntm, ntn, hb, wb, bal)) {
    const int l = x * ((ntm * ntn + 7) / 8) + idx;
    run_tile(l, ntm, ntn, tm, tn);
    return l < ntm * ntn;
  }
  const int bi = x / bn, bj = x % bn;
  tm = bi * hb + idx / wb;
  tn = bj * wb + idx % wb;
  return tm < ntm && tn < ntn;
}
template <int SBM, int SBN>
__device__ __forceinline__ bool tile_of_block_g(int bid, int grid, int ntm, int ntn, int& tm, int& tn, bool bal = false);
template <int SBM, int SBN>
__device__ __forceinline__ bool tile_of_block(int bid, int ntm, int ntn, int& tm, int& tn, bool bal = false) {
  return tile_of_block_g<SBM, SBN>(bid, (int)gridDim.x, ntm, ntn, tm, tn, bal);
}
// the same with the size of the (virtual) grid given: persistent workgroups walk a grid larger than the launch
template <int SBM, int SBN>
__device__ __forceinline__ bool tile_of_block_g(int bid, int grid, int ntm, int ntn, int& tm, int& tn, bool bal) {
  if (grid == xcd_grid(ntm, ntn, bal)) {
    // small problem (fewer than 4 super-blocks per XCD): one rectangular block of tiles per XCD, so an L2 only
    // streams the operand bands of its block.  (Plain round-robin numbering gave each XCD one tile COLUMN: all of
    // A streamed into every L2, 8x the bytes in the PMC counters.)
    return xcd_tile(bid, ntm, ntn, tm, tn, bal);
  }
  const int x = bid & 7, i = bid >> 3;
  const int per = SBM * SBN;
  const int sb = (i / per) * 8 + x, t = i % per;
  const int nsbn = (ntn + SBN - 1) / SBN;
  const int sbm = sb / nsbn, sbn = sb % nsbn;
  tm = sbm * SBM + (t % SBM);
  tn = sbn * SBN + (t / SBM);
  return tm < ntm && tn < ntn;
}
template <int SBM, int SBN>
static inline unsigned tile_grid(int ntm, int ntn, bool bal = false) {
  if (ntm * ntn < 4 * 8 * SBM * SBN) return (unsigned)xcd_grid(ntm, ntn, bal);  // fewer than 4 super-blocks per XCD
  const int nsb = ((ntm + SBM - 1) / SBM) * ((ntn + SBN - 1) / SBN);
  return (unsigned)(((nsb + 7) / 8) * 8 * SBM * SBN);
}

// ---- epilogue --------------------------------------------------------------------------------
// Each wave parks its 64x64 fp32 tile in LDS (row pitch 68 floats: conflict-free b128 writes),
// then walks it 4 rows x 64 columns at a time in a rolled loop so the generic (runtime-selected)
// epilogue is emitted once and every row is stored as one contiguous 128-B (bf16) / 256-B (fp32)
// segment.  mw/nw: global row/column of the wave's tile origin.
// the 8 bias values of a lane's columns (nw: the wave tile's first column); requested early by the kernels that can
// spare the registers, so that the memory round trip is over when the epilogue starts
__device__ __forceinline__ void epi_load_bias(const CaGemmDesc& d, int lane, int nw, int z1, int z2, float (&bias8)[8]) {
  const int nb = nw + 8 * (lane & 7);
  const int nvalid = (d.N - nb) < 8 ? (d.N - nb) : 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if (d.bias && nvalid > 0) {
    const float* bz = d.bias + z1 * d.sBias1 + z2 * d.sBias2 + nb;
    if (nvalid == 8 && (((uintptr_t)bz) & 15) == 0) {
      const f32x4_t b0 = *(const f32x4_t*)bz, b1 = *(const f32x4_t*)(bz + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) bias8[e] = e < 4 ? b0[e] : b1[e - 4];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < nvalid) bias8[e] = bz[e];
    }
  }
}
// CaGemmDesc.C8: eight consecutive activations as e4m3 with the delayed per-tensor scale; the lane's running max|v|
__device__ __forceinline__ void ca_store_fp8x8(unsigned char* dst, const float (&v)[8], float scale, float& amx) {
  float t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    amx = fmaxf(amx, fabsf(v[e]));
    t[e] = fminf(fmaxf(v[e] * scale, -448.0f), 448.0f);
  }
  unsigned int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], w1, true);
  *(uint2*)dst = make_uint2(w0, w1);
}

// ---- fast epilogue: a 64 x 64 wave tile that lies wholly inside the output, 16-byte aligned rows -------------------
// The general walk below decides everything per lane and per pass at run time (ragged columns, rows beyond M, which
// epilogue, unaligned dropout groups, both output types): ~400 vector instructions per pass of 8 elements per lane in
// the ISA, against ~130 of arithmetic - on the FFN GEMMs the epilogue's instruction count, not the stores' bytes, is
// what the tile waits for (3.3 VALU per MFMA over the whole kernel).  Interior wave tiles (all but the last row / column
// of tiles) take this form instead: the epilogue kind, the output type and dropout are template parameters, all
// predicates are gone, the flat element index of the dropout hash advances by a constant.  Same arithmetic in the same
// order as the general walk: the results are bit-identical (tests/test_kernels_gpu.py compares ragged and interior tiles
// of one launch against the same reference).
// ---- slab staging (kernel X, round 5) ---------------------------------------------------------------------------
// The classic staging parks a wave's whole 64 x 64 fp32 tile (17 KB, 139 KB for the 8 waves of kernel X: every byte of
// the operand stages), so nothing of the next tile can be in flight while a tile leaves.  Slab staging parks 16 rows
// at a time in a 4-KiB region per wave that lies BEHIND the two operand stages (128 KiB + 8 x 4 KiB = all 160 KiB of
// LDS), walks them in the same two passes of 8 rows x 64 columns per lane group as before (same lane -> element
// assignment, same order: bit-identical outputs and sums of squares) and lets kernel X request the next tile's first
// K-step before the epilogue starts.  The 16 x 64 slab has no row padding: 16-byte chunk c of row r sits at chunk
// c ^ r, which is conflict-free for the parked fragments (8 consecutive rows of one chunk column per ds_write_b128
// group) and for the row-major reads (the hardware's 16-lane ds_read_b128 groups cover 16 different chunks).
// Which 16 rows are parked is a run-time choice in a rolled loop - made by a wave-uniform switch over register
// copies, because a run-time index into the accumulator array would send it through scratch memory.
#define SLAB_BYTES 4096
__device__ __forceinline__ void slab_park(const f32x4_t (&acc)[4][4], int i4, float* slab, int lane) {
  const int r = lane & 15, g = lane >> 4;
  float* row = slab + r * 64;
  float* p0 = row + ((g ^ r) << 2);
  float* p1 = row + (((4 + g) ^ r) << 2);
  float* p2 = row + (((8 + g) ^ r) << 2);
  float* p3 = row + (((12 + g) ^ r) << 2);
  // the stores sit INSIDE the cases: straight from the accumulator registers, no copies (the asm statements keep the
  // cases as branches: as selects they would cost 3 x 16 v_cndmask per slab)
#define SLAB_CASE(I)                 \
  asm volatile("; slab rows " #I);   \
  *(f32x4_t*)p0 = acc[I][0];         \
  *(f32x4_t*)p1 = acc[I][1];         \
  *(f32x4_t*)p2 = acc[I][2];         \
  *(f32x4_t*)p3 = acc[I][3];
  switch (i4) {
    case 0: SLAB_CASE(0) break;
    case 1: SLAB_CASE(1) break;
    case 2: SLAB_CASE(2) break;
    default: SLAB_CASE(3) break;
  }
#undef SLAB_CASE
}
// the 8 consecutive columns 8 (lane & 7) .. of row p * 8 + (lane >> 3) of the parked slab
__device__ __forceinline__ void slab_read(const float* slab, int p, int lane, f32x4_t& a4, f32x4_t& b4) {
  const int r = p * 8 + (lane >> 3), c = 2 * (lane & 7);
  const float* row = slab + r * 64;
  a4 = *(const f32x4_t*)(row + ((c ^ r) << 2));
  b4 = *(const f32x4_t*)(row + (((c + 1) ^ r) << 2));
}

template <int EPI, bool F32, bool DROP, bool SLAB = false>
__device__ __forceinline__ void gemm_epilogue_fast(const CaGemmDesc& d, const float* wt, int lane, int mw, int nb, int z,
                                                   int64_t zoffC, int64_t zoffR, const float (&bias8)[8], float& ssq,
                                                   float& amx, const f32x4_t (*acc)[4][4] = nullptr) {
  const int M = d.M, N = d.N;
  const float alpha = d.alpha;
  const float keep_scale = DROP ? 1.f / (1.f - d.dropout_p) : 1.f;
  const int r0 = lane >> 3, c0 = 8 * (lane & 7);
  constexpr bool NEEDS_R = EPI == CA_EPI_RESIDUAL || EPI == CA_EPI_DGELU;
  const unsigned short* Rp = NEEDS_R ? (const unsigned short*)d.R + zoffR + nb + (int64_t)(mw + r0) * d.ldr : nullptr;
  const int64_t rstep = 8 * d.ldr;
  int64_t coff = zoffC + (int64_t)(mw + r0) * d.ldc + nb;
  const int64_t cstep = 8 * d.ldc;
  uint64_t idx = ((uint64_t)z * M + (mw + r0)) * (uint64_t)N + nb;  // multiple of 4: N % 8 == 0, nb % 8 == 0
  const uint64_t istep = 8ull * (uint64_t)N;
  u16x8_t r_next = {0, 0, 0, 0, 0, 0, 0, 0};
  if (NEEDS_R) r_next = *(const u16x8_t*)Rp;
  const bool c8_on = (EPI == CA_EPI_GELU || EPI == CA_EPI_DGELU) && d.C8 != nullptr;  // (wave-uniform)
  const float s8 = c8_on ? d.c8_scale[0] : 1.f;
  // dropout: the hash takes the group index (flat element index / 4) as two 32-bit words.  Where the wave tile does
  // not cross a multiple of 2^34 elements (wave-uniform test) the high word is a constant folded into the seed word
  // and the low word advances by a 32-bit add per pass - the same bits as ca_dropout_keep4 on the 64-bit index.
  // (the caller sends a wave tile that does cross such a multiple through the general walk: epi_drop32_ok)
  unsigned int dg = 0, ds_eff = 0, dthr = 0;
  if (DROP) {
    const uint64_t ilo = ((uint64_t)z * M + mw) * (uint64_t)N;
    const unsigned int sd = (unsigned int)d.dropout_seed * 0x9E3779B9u + (unsigned int)(d.dropout_seed >> 32);
    ds_eff = sd ^ ((unsigned int)(ilo >> 34) * 0x85EBCA6Bu);
    dg = (unsigned int)(idx >> 2);
    dthr = ca_dropout_threshold(d.dropout_p);
  }
  const unsigned int dgstep = (unsigned int)(istep >> 2);
  if (SLAB) slab_park(*acc, 0, const_cast<float*>(wt), lane);
#pragma unroll 1
  for (int i4 = 0; i4 < (SLAB ? 4 : 1); ++i4) {
  // slab i4's 16 rows go to registers first, then the next 16 rows are parked (LDS operations of a wave execute in
  // order: the reads see the old slab) - the stores complete under this slab's arithmetic
  f32x4_t sa[2], sb[2];
  if (SLAB) {
    slab_read(wt, 0, lane, sa[0], sb[0]);
    slab_read(wt, 1, lane, sa[1], sb[1]);
    if (i4 < 3) slab_park(*acc, i4 + 1, const_cast<float*>(wt), lane);
  }
#pragma unroll 2
  for (int itl = 0; itl < (SLAB ? 2 : 8); ++itl) {
    const int it = SLAB ? 2 * i4 + itl : itl;
    const u16x8_t r_cur = r_next;
    if (NEEDS_R && it < 7) r_next = *(const u16x8_t*)(Rp + (it + 1) * rstep);
    f32x4_t a4, b4;
    if (SLAB) {
      a4 = sa[itl];
      b4 = sb[itl];
    } else {
      const float* wr = wt + (it * 8 + r0) * EPI_PITCH + c0;
      a4 = *(const f32x4_t*)wr;
      b4 = *(const f32x4_t*)(wr + 4);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? a4[e] : b4[e - 4]) * alpha + bias8[e];
    unsigned int keep = 0xFFu;
    if (DROP) {
      keep = ca_dropout_keep4_lo(ds_eff, dg, dthr) | (ca_dropout_keep4_lo(ds_eff, dg + 1, dthr) << 4);
      dg += dgstep;
    }
    float v2[8];
    if (EPI == CA_EPI_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = gelu_erf(v[e]);
        if (DROP) g = ((keep >> e) & 1u) ? g * keep_scale : 0.f;
        v2[e] = g + 0.f;  // (the general walk adds the residual slot, zero here: -0 becomes +0 there too)
      }
    } else if (EPI == CA_EPI_RESIDUAL) {
      if (DROP) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * keep_scale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(r_cur[e]);
    } else if (EPI == CA_EPI_DGELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float dg = dgelu_erf(bf2f(r_cur[e]));
        if (DROP) dg = ((keep >> e) & 1u) ? dg * keep_scale : 0.f;
        v[e] *= dg;
      }
    }
    if (F32) {
      float* C = (float*)d.C + coff;
      if (d.accumulate) {
        const f32x4_t c0v = *(const f32x4_t*)C, c1v = *(const f32x4_t*)(C + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += e < 4 ? c0v[e] : c1v[e - 4];
      }
      if (d.c_stream_out) {
        __builtin_nontemporal_store((f32x4_t){v[0], v[1], v[2], v[3]}, (f32x4_t*)C);
        __builtin_nontemporal_store((f32x4_t){v[4], v[5], v[6], v[7]}, (f32x4_t*)(C + 4));
      } else {
        *(f32x4_t*)C = (f32x4_t){v[0], v[1], v[2], v[3]};
        *(f32x4_t*)(C + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) ssq = fmaf(v[e], v[e], ssq);
    } else {
      u16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
      if (d.c_stream_out)
        __builtin_nontemporal_store(o, (u16x8_t*)((unsigned short*)d.C + coff));
      else
        *(u16x8_t*)((unsigned short*)d.C + coff) = o;
      if (EPI == CA_EPI_NONE && d.c_sumsq != nullptr) {  // (wave-uniform) weight gradients kept in bf16: the norm of what is stored
#pragma unroll
        for (int e = 0; e < 8; ++e) ssq = fmaf(bf2f(o[e]), bf2f(o[e]), ssq);
      }
      if (EPI == CA_EPI_DGELU && c8_on) ca_store_fp8x8((unsigned char*)d.C8 + coff, v, s8, amx);
    }
    if (EPI == CA_EPI_GELU) {
      u16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v2[e]);
      *(u16x8_t*)((unsigned short*)d.C2 + coff) = o;
      if (c8_on) ca_store_fp8x8((unsigned char*)d.C8 + coff, v2, s8, amx);
    }
    coff += cstep;
  }
  }  // slabs
}

// PARKED: the 64 x 64 staging tile `wave` already holds the accumulators (kernel M: two waves fill one tile)
// SLAB: `smem` is the wave's own 4-KiB slab (see slab_park); the 64 x 64 tile goes through it 16 rows at a time
template <bool PARKED = false, bool SLAB = false>
__device__ __forceinline__ void gemm_epilogue(const CaGemmDesc& d, f32x4_t (&acc)[4][4], char* smem,
                                              int wave, int lane, int mw, int nw, int z, int z1,
                                              int z2, const float (*bias_pre)[8] = nullptr) {
  // lane -> 8 consecutive columns of one row; 8 rows per pass, 8 passes; 16-byte bf16 stores
  const int M = d.M, N = d.N;
  const int64_t zoffC = z1 * d.sC1 + z2 * d.sC2;
  const int64_t zoffR = z1 * d.sR1 + z2 * d.sR2;
  const bool vec_ok = ((d.ldc & 7) == 0) && ((zoffC & 7) == 0) && ((d.ldr & 7) == 0) && ((zoffR & 7) == 0);
  const float keep_scale = d.dropout_p > 0.f ? 1.f / (1.f - d.dropout_p) : 1.f;
  const int nb = nw + 8 * (lane & 7);
  const int nvalid = (N - nb) < 8 ? (N - nb) : 8;
  const bool full = nvalid == 8 && vec_ok;
  const int epi = d.epilogue;
  const bool has_gelu = epi == CA_EPI_GELU || epi == CA_EPI_GELU_RESIDUAL;
  const bool needs_r = epi == CA_EPI_RESIDUAL || epi == CA_EPI_DGELU || epi == CA_EPI_GELU_RESIDUAL;
  // Everything the epilogue reads from global memory is requested BEFORE the accumulators go through LDS, and the R
  // row of pass it + 1 before pass it is computed: requested where they were used, the bias (8 dword loads per lane)
  // and each pass's R row put one exposed memory round trip each in front of the stores (+11 us for the bias and
  // +5..10 us for R on a 120-us launch at [3992 x 7680 x 1920]).
  float bias8[8];
  if (bias_pre) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (*bias_pre)[e];
  } else {
    epi_load_bias(d, lane, nw, z1, z2, bias8);
  }
  const unsigned short* Rbase = (const unsigned short*)d.R + zoffR + nb;
  auto load_r = [&](int it, u16x8_t& u) {  // R row of pass `it` (fast form only; ragged columns load in the pass)
    const int m = mw + it * 8 + (lane >> 3);
    if (needs_r && full && it < 8 && m < M) u = *(const u16x8_t*)(Rbase + (int64_t)m * d.ldr);
  };
  u16x8_t r_next = {0, 0, 0, 0, 0, 0, 0, 0};
  load_r(0, r_next);

  float* wt = SLAB ? (float*)smem : (float*)smem + wave * (64 * EPI_PITCH);
  if (!PARKED && !SLAB) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(f32x4_t*)(wt + (i * 16 + (lane & 15)) * EPI_PITCH + j * 16 + 4 * (lane >> 4)) = acc[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier
  }
  const bool c8_on = d.C8 != nullptr && (epi == CA_EPI_GELU || epi == CA_EPI_DGELU);
  // (slab staging: every lane keeps parking its share of the rows inside the walk, also one without valid columns)
  if (!SLAB && nvalid <= 0 && d.c_sumsq == nullptr && !c8_on) return;
  float ssq = 0.f;  // sum of squares of the fp32 values this lane stores (c_sumsq)
  float amx = 0.f;  // max |gelu| this lane stores (C8)
  const float s8 = c8_on ? d.c8_scale[0] : 1.f;
  // interior wave tile (wave-uniform test): the specialised walk above
  const bool drop_on = d.dropout_p > 0.f;
  // (dropout in the specialised walk: the 64 rows' flat element indices share their bits from 34 up)
  const bool drop32_ok = !drop_on || ((((uint64_t)z * M + mw) * (uint64_t)N) >> 34) == ((((uint64_t)z * M + mw + 64) * (uint64_t)N) >> 34);
  const bool interior = g_ca_epi_general == 0 && mw + 64 <= M && nw + 64 <= N && vec_ok && (N & 7) == 0 && d.C != nullptr && drop32_ok &&
                        (d.out_f32 || !d.accumulate) && (!has_gelu || (epi == CA_EPI_GELU && d.C2 != nullptr && !d.out_f32)) &&
                        !(d.out_f32 && epi != CA_EPI_NONE) && !(drop_on && epi == CA_EPI_NONE);
  if (interior) {
#define EPI_FAST(E, F, D) gemm_epilogue_fast<E, F, D, SLAB>(d, wt, lane, mw, nb, z, zoffC, zoffR, bias8, ssq, amx, &acc)
    if (d.out_f32) {
      EPI_FAST(CA_EPI_NONE, true, false);
    } else if (epi == CA_EPI_NONE) {
      EPI_FAST(CA_EPI_NONE, false, false);
    } else if (epi == CA_EPI_GELU) {
      if (drop_on) EPI_FAST(CA_EPI_GELU, false, true); else EPI_FAST(CA_EPI_GELU, false, false);
    } else if (epi == CA_EPI_RESIDUAL) {
      if (drop_on) EPI_FAST(CA_EPI_RESIDUAL, false, true); else EPI_FAST(CA_EPI_RESIDUAL, false, false);
    } else {
      if (drop_on) EPI_FAST(CA_EPI_DGELU, false, true); else EPI_FAST(CA_EPI_DGELU, false, false);
    }
#undef EPI_FAST
  } else {
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    if (SLAB && (it & 1) == 0) slab_park(acc, it >> 1, wt, lane);  // (before any `continue`: every lane parks its rows)
    const int ml = it * 8 + (lane >> 3);
    const int m = mw + ml;
    const u16x8_t r_cur = r_next;
    load_r(it + 1, r_next);
    if (m >= M || nvalid <= 0) continue;
    f32x4_t a4, b4;
    if (SLAB) {
      slab_read(wt, it & 1, lane, a4, b4);
    } else {
      a4 = *(const f32x4_t*)(wt + ml * EPI_PITCH + 8 * (lane & 7));
      b4 = *(const f32x4_t*)(wt + ml * EPI_PITCH + 8 * (lane & 7) + 4);
    }
    float v[8], v2[8], r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = (e < 4 ? a4[e] : b4[e - 4]) * d.alpha + bias8[e];
      v2[e] = 0.f;
      r[e] = 0.f;
    }
    const int64_t coff = zoffC + (int64_t)m * d.ldc + nb;
    if (needs_r) {
      if (full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = bf2f(r_cur[e]);
      } else {
        const unsigned short* R = Rbase + (int64_t)m * d.ldr;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nvalid) r[e] = bf2f(R[e]);
      }
    }
    unsigned int keep = 0xFFu;
    if (d.dropout_p > 0.f && (has_gelu || epi == CA_EPI_DGELU || epi == CA_EPI_RESIDUAL)) {
      const uint64_t idx = ((uint64_t)z * M + m) * (uint64_t)N + nb;
      if ((idx & 3) == 0) {  // aligned group: two hashes for the 8 elements
        keep = ca_dropout_keep4(d.dropout_seed, idx, d.dropout_p) |
               (ca_dropout_keep4(d.dropout_seed, idx + 4, d.dropout_p) << 4);
      } else {
        keep = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) keep |= (ca_dropout_keep(d.dropout_seed, idx + e, d.dropout_p) ? 1u : 0u) << e;
      }
    }
    if (has_gelu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = gelu_erf(v[e]);
        if (d.dropout_p > 0.f) g = ((keep >> e) & 1u) ? g * keep_scale : 0.f;
        v2[e] = g + r[e];  // r is zero unless GELU_RESIDUAL
      }
    } else if (epi == CA_EPI_RESIDUAL) {
      // C = R + dropout(alpha A.B + bias): hidden-state dropout on the sub-layer output before the residual add
      if (d.dropout_p > 0.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * keep_scale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    } else if (epi == CA_EPI_DGELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float dg = dgelu_erf(r[e]);
        if (d.dropout_p > 0.f) dg = ((keep >> e) & 1u) ? dg * keep_scale : 0.f;
        v[e] *= dg;
      }
    }
    if (d.out_f32) {
      float* C = (float*)d.C + coff;
      if (full) {
        if (d.accumulate) {
          const f32x4_t c0 = *(const f32x4_t*)C, c1 = *(const f32x4_t*)(C + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += e < 4 ? c0[e] : c1[e - 4];
        }
        *(f32x4_t*)C = (f32x4_t){v[0], v[1], v[2], v[3]};
        *(f32x4_t*)(C + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
#pragma unroll
        for (int e = 0; e < 8; ++e) ssq = fmaf(v[e], v[e], ssq);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nvalid) {
            const float t = d.accumulate ? C[e] + v[e] : v[e];
            C[e] = t;
            ssq = fmaf(t, t, ssq);
          }
      }
    } else if (d.C) {
      unsigned short* C = (unsigned short*)d.C + coff;
      if (full) {
        if (d.accumulate) {
          const u16x8_t c = *(const u16x8_t*)C;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(c[e]);
        }
        u16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        *(u16x8_t*)C = o;
        if (d.c_sumsq != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ssq = fmaf(bf2f(o[e]), bf2f(o[e]), ssq);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nvalid) {
            const unsigned short t = f2bf(d.accumulate ? bf2f(C[e]) + v[e] : v[e]);
            C[e] = t;
            ssq = fmaf(bf2f(t), bf2f(t), ssq);
          }
      }
    }
    if (has_gelu && d.C2) {
      unsigned short* C2 = (unsigned short*)d.C2 + coff;
      if (full) {
        u16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v2[e]);
        *(u16x8_t*)C2 = o;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nvalid) C2[e] = f2bf(v2[e]);
      }
    }
    if (c8_on) {
      unsigned char* C8 = (unsigned char*)d.C8 + coff;
      float w8[8];  // the activation (GELU) or the gradient (GELU') this tile just formed
#pragma unroll
      for (int e = 0; e < 8; ++e) w8[e] = epi == CA_EPI_GELU ? v2[e] : v[e];
      if (full) {
        ca_store_fp8x8(C8, w8, s8, amx);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nvalid) {
            amx = fmaxf(amx, fabsf(w8[e]));
            const float t = fminf(fmaxf(w8[e] * s8, -448.0f), 448.0f);
            C8[e] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(t, 0.f, 0u, false) & 0xffu);
          }
      }
    }
  }
  }  // general walk
  if (d.c_sumsq != nullptr) {
    // (every lane of the wave arrives here: fixed butterfly order, the same bits on every run)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o, 64);
    if (lane == 0 && mw < M && nw < N) d.c_sumsq[(int64_t)(mw >> 6) * ((N + 63) >> 6) + (nw >> 6)] = ssq;
  }
  if (c8_on && d.c8_amax != nullptr) {
    // (every lane of the wave arrives here; a maximum does not depend on the order: the same bits on every run)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
    // (spread over the accumulator's words: the 15 000 wave tiles of a 12 000 x 5 120 output on one address took 200 us)
    if (lane == 0 && amx > 0.f)
      atomicMax((unsigned int*)d.c8_amax + (((mw >> 6) * 7 + (nw >> 6)) & (CA_FP8_AMAX_SLOTS - 1)), __float_as_uint(amx));
  }
}

// ---- kernel S: 128x128 tile, 4 waves, 2 LDS stages, 2 blocks/CU (small / batched problems) ----
template <int AL, int BL, bool KS>
__global__ __launch_bounds__(256) void ca_gemm_kernel(const CaGemmDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  if (!tile_of_block<8, 8>(blockIdx.x, (d.M + BM - 1) / BM, (d.N + BN - 1) / BN, tm, tn, d.xcd_balanced != 0)) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z1 = z / d.batch2, z2 = z % d.batch2;

  const __bf16* A = (const __bf16*)d.A + z1 * d.sA1 + z2 * d.sA2;
  const __bf16* B = (const __bf16*)d.B + z1 * d.sB1 + z2 * d.sB2;

  // (loaders and pipeline as in kernel X below, at half the tile: see the comments there)
  KMajorStream<4> la_k, lb_k;
  MNMajorStream<4, 16> la_f, lb_f;
  MNMajorLoader<4, 16, KS> la_m, lb_m;
  if (AL == CA_KMAJOR)
    la_k.init(A, d.lda, m0, d.M, wave, lane);
  else if (KS)
    la_m.init(A, d.lda, d.a_kseg, d.a_kseg_stride, m0, d.M, wave, lane);
  else
    la_f.init(A, d.lda, m0, d.M, wave, lane);
  if (BL == CA_KMAJOR)
    lb_k.init(B, d.ldb, n0, d.N, wave, lane);
  else if (KS)
    lb_m.init(B, d.ldb, d.b_kseg, d.b_kseg_stride, n0, d.N, wave, lane);
  else
    lb_f.init(B, d.ldb, n0, d.N, wave, lane);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int K = d.K;
  const int nk = (K + BK - 1) / BK;

  // LDS: A0 | A1 | B0 | B1 (two stages of each 16-KiB operand tile).  A tile's 8 LDS-DMA per wave go out as one burst
  // behind the barrier, a full K-step before the tile is needed.
  auto burst = [&](int kt) {
    if (kt >= nk) return;
    char* na = smem + (kt & 1) * TILE_BYTES;
    char* nb = na + 2 * TILE_BYTES;
    const bool full = (kt + 1) * BK <= K;  // wave-uniform
    if (full) {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        if (AL == CA_MNMAJOR && KS)
          la_m.issue_one(na, wave, lane, kt * BK, K, part);
        else if (AL == CA_KMAJOR)
          la_k.issue_one(na, wave, part);
        else
          la_f.issue_one(na, wave, part);
        if (BL == CA_MNMAJOR && KS)
          lb_m.issue_one(nb, wave, lane, kt * BK, K, part);
        else if (BL == CA_KMAJOR)
          lb_k.issue_one(nb, wave, part);
        else
          lb_f.issue_one(nb, wave, part);
      }
    } else {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        if (AL == CA_MNMAJOR && KS)
          la_m.issue_one(na, wave, lane, kt * BK, K, part);
        else if (AL == CA_KMAJOR)
          la_k.issue_one_tail(na, wave, K - kt * BK, part);
        else
          la_f.issue_one_tail(na, wave, lane, K - kt * BK, part);
        if (BL == CA_MNMAJOR && KS)
          lb_m.issue_one(nb, wave, lane, kt * BK, K, part);
        else if (BL == CA_KMAJOR)
          lb_k.issue_one_tail(nb, wave, K - kt * BK, part);
        else
          lb_f.issue_one_tail(nb, wave, lane, K - kt * BK, part);
      }
    }
    if (AL == CA_KMAJOR) la_k.advance();
    if (AL == CA_MNMAJOR && !KS) la_f.advance();
    if (BL == CA_KMAJOR) lb_k.advance();
    if (BL == CA_MNMAJOR && !KS) lb_f.advance();
  };
  auto zero_tail = [&](int kt) {
    if (kt != nk - 1 || nk * BK == K) return;
    char* na = smem + (kt & 1) * TILE_BYTES;
    char* nb = na + 2 * TILE_BYTES;
    const int krem = K - kt * BK;
    if (AL == CA_KMAJOR) la_k.zero_fix(na, wave, lane, krem);
    if (AL == CA_MNMAJOR && !KS) la_f.zero_fix(na, wave, lane, krem);
    if (BL == CA_KMAJOR) lb_k.zero_fix(nb, wave, lane, krem);
    if (BL == CA_MNMAJOR && !KS) lb_f.zero_fix(nb, wave, lane, krem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // One barrier per K-step: a wave arrives at barrier(kt) only after it consumed every fragment of
  // tile kt-1 (lgkmcnt(0) below), so the stage of tile kt-1 may be re-staged right after the barrier.
  // (Two workgroups share a CU and fall into complementary phases by themselves - one reads while the other
  // multiplies; kernel X's finer interleaving of reads and MFMAs was measured 5-10 % slower here.)
  burst(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tile kt landed; my reads of kt-1 done
    zero_tail(kt);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    burst(kt + 1);  // (splitting this burst across the MFMA blocks was measured slower here)

    const char* ta = smem + (kt & 1) * TILE_BYTES;
    const char* tb = ta + 2 * TILE_BYTES;
    // two k-halves per K-step; the fragments of the second half are read before the MFMAs of the first
    auto read_frags = [&](int s, bf16x8_t (&af)[4], bf16x8_t (&bf)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = (AL == CA_KMAJOR) ? frag_kmajor_async(ta, wm * 64 + i * 16, s, lane)
                                  : frag_mnmajor<256>(ta, wm * 64 + i * 16, s, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[j] = (BL == CA_KMAJOR) ? frag_kmajor_async(tb, wn * 64 + j * 16, s, lane)
                                  : frag_mnmajor<256>(tb, wn * 64 + j * 16, s, lane);
    };
    auto mma = [&](bf16x8_t (&af)[4], bf16x8_t (&bf)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    };
    bf16x8_t a0[4], b0[4], a1[4], b1[4];
    read_frags(0, a0, b0);
    lds_wait(a0);
    lds_wait(b0);
    read_frags(1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    lds_wait(a1);
    lds_wait(b1);
    mma(a1, b1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all waves are done with the stages before the epilogue reuses the LDS
  asm volatile("" ::: "memory");
  gemm_epilogue(d, acc, smem, wave, lane, m0 + wm * 64, n0 + wn * 64, z, z1, z2);
}

// ---- kernel M: 128x128 tile, 8 waves (2x4, 64x32 each), 4 LDS stages, one workgroup per CU --------------------------
// For grids of at most one 128x128 tile per CU (the N = d GEMMs of the d = 1024 models at M = 3992: 256 tiles; the
// Whisper decoder's teacher-forced rows: 56-64 tiles).  There kernel S runs a lone wave per SIMD through
// barrier -> fragment reads -> 32 MFMAs -> ..., 1 400-1 700 cycles per K-step for 512 of MFMA whatever the ring depth
// (tools/archive/dev_dec_gemm.py), and kernel L's 256x128 tiles use half the CUs.  Here the same tile is shared by two waves
// per SIMD (64 rows x 32 columns each, 16 MFMAs per K-step and wave): one wave's fragment reads and barrier wait run
// under the other's MFMAs.  The LDS one workgroup per CU leaves free holds a ring of four stages (LDS-DMA three tiles
// ahead, counted vmcnt: 4 pieces per wave and tile).  Epilogue: the two waves of a 64x64 quadrant park their halves in
// one staging tile, the even one walks it through the shared epilogue.
#define M_NST 4
#define M_LDS_BYTES (2 * M_NST * TILE_BYTES)  // 131072 >= 4 quadrants * 64*68*4 (69632) of epilogue staging
template <int N>
__device__ __forceinline__ void lds_wait_n(bf16x8_t (&f)[N]) {
  if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
}
template <int AL, int BL>
__global__ __launch_bounds__(512) void ca_gemm_kernel_m(const CaGemmDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, 64 (M) x 32 (N) each
  int tm, tn;
  if (!tile_of_block<8, 8>(blockIdx.x, (d.M + BM - 1) / BM, (d.N + BN - 1) / BN, tm, tn, d.xcd_balanced != 0)) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z1 = z / d.batch2, z2 = z % d.batch2;
  const __bf16* A = (const __bf16*)d.A + z1 * d.sA1 + z2 * d.sA2;
  const __bf16* B = (const __bf16*)d.B + z1 * d.sB1 + z2 * d.sB2;
  const int K = d.K;
  const int nk = (K + BK - 1) / BK;

  KMajorStream<2> la_k, lb_k;
  MNMajorStream<2, 16> la_f, lb_f;
  if (AL == CA_KMAJOR) la_k.init(A, d.lda, m0, d.M, wave, lane); else la_f.init(A, d.lda, m0, d.M, wave, lane);
  if (BL == CA_KMAJOR) lb_k.init(B, d.ldb, n0, d.N, wave, lane); else lb_f.init(B, d.ldb, n0, d.N, wave, lane);

  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // LDS: A0 .. A3 | B0 .. B3 (16 KiB each)
  auto burst = [&](int kt) {  // this wave's share of tile kt: 2 A pieces + 2 B pieces
    if (kt >= nk) return;
    char* na = smem + (kt % M_NST) * TILE_BYTES;
    char* nb = na + M_NST * TILE_BYTES;
    const bool full = (kt + 1) * BK <= K;  // wave-uniform
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      if (full) {
        if (AL == CA_KMAJOR) la_k.issue_one(na, wave, part); else la_f.issue_one(na, wave, part);
        if (BL == CA_KMAJOR) lb_k.issue_one(nb, wave, part); else lb_f.issue_one(nb, wave, part);
      } else {
        if (AL == CA_KMAJOR) la_k.issue_one_tail(na, wave, K - kt * BK, part); else la_f.issue_one_tail(na, wave, lane, K - kt * BK, part);
        if (BL == CA_KMAJOR) lb_k.issue_one_tail(nb, wave, K - kt * BK, part); else lb_f.issue_one_tail(nb, wave, lane, K - kt * BK, part);
      }
    }
    if (AL == CA_KMAJOR) la_k.advance(); else la_f.advance();
    if (BL == CA_KMAJOR) lb_k.advance(); else lb_f.advance();
  };
  auto zero_tail = [&](int kt) {
    if (kt != nk - 1 || nk * BK == K) return;
    char* na = smem + (kt % M_NST) * TILE_BYTES;
    char* nb = na + M_NST * TILE_BYTES;
    const int krem = K - kt * BK;
    if (AL == CA_KMAJOR) la_k.zero_fix(na, wave, lane, krem); else la_f.zero_fix(na, wave, lane, krem);
    if (BL == CA_KMAJOR) lb_k.zero_fix(nb, wave, lane, krem); else lb_f.zero_fix(nb, wave, lane, krem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int p = 0; p < M_NST - 1; ++p) burst(p);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (this wave's pieces; the barrier covers the others); tiles kt+1, kt+2 may still be in flight
    if (kt + 2 < nk)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (kt + 1 < nk)
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    zero_tail(kt);
    __builtin_amdgcn_s_barrier();  // every wave has consumed tile kt-1: its stage is free
    asm volatile("" ::: "memory");
    burst(kt + M_NST - 1);

    const char* ta = smem + (kt % M_NST) * TILE_BYTES;
    const char* tb = ta + M_NST * TILE_BYTES;
    auto read_frags = [&](int sh, bf16x8_t (&af)[4], bf16x8_t (&bf)[2]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = (AL == CA_KMAJOR) ? frag_kmajor_async(ta, wm * 64 + i * 16, sh, lane)
                                  : frag_mnmajor<256>(ta, wm * 64 + i * 16, sh, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bf[j] = (BL == CA_KMAJOR) ? frag_kmajor_async(tb, wn * 32 + j * 16, sh, lane)
                                  : frag_mnmajor<256>(tb, wn * 32 + j * 16, sh, lane);
    };
    auto mma = [&](bf16x8_t (&af)[4], bf16x8_t (&bf)[2]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    };
    bf16x8_t a0[4], b0[2], a1[4], b1[2];
    read_frags(0, a0, b0);
    lds_wait_n(a0);
    lds_wait_n(b0);
    read_frags(1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    lds_wait_n(a1);
    lds_wait_n(b1);
    mma(a1, b1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all waves are done with the stages before the epilogue reuses the LDS
  asm volatile("" ::: "memory");
  // quadrant (wm, wn >> 1): this wave's 64 x 32 half goes to columns (wn & 1) * 32 of the quadrant's staging tile
  const int quad = wm * 2 + (wn >> 1);
  float* wt = (float*)smem + quad * (64 * EPI_PITCH);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *(f32x4_t*)(wt + (i * 16 + (lane & 15)) * EPI_PITCH + (wn & 1) * 32 + j * 16 + 4 * (lane >> 4)) = acc[i][j];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if ((wn & 1) == 0) {
    f32x4_t none[4][4];
    gemm_epilogue<true>(d, none, smem, quad, lane, m0 + wm * 64, n0 + (wn >> 1) * 64, z, z1, z2);
  }
}

// ---- kernel X: 256x256 tile, 8 waves (2x4, 128x64 each), 2 LDS stages, one workgroup per CU ----------
// Half the LDS-fill bytes per FLOP of kernel S (the fill stream is what bounds S);
// used when both output dimensions are large enough to give every CU a tile.
#define XBM 256
#define XBN 256
#define XTILE (XBM * BK * 2)  // 32 KiB per operand tile
#define XSTAGE (2 * XTILE)
#define X_LDS_BYTES (8 * 64 * EPI_PITCH * 4)  // 139264 >= 2 stages * 64 KiB (classic staging: the fp8 256x256 kernel)
#define X_SLAB_BASE (2 * XSTAGE)             // bf16 kernel X: 8 epilogue slabs of 4 KiB behind the two stages
#define X_SLAB_LDS (X_SLAB_BASE + 8 * SLAB_BYTES)  // 163840 = all 160 KiB of a CU's LDS
#ifndef CA_X_SLAB
// 1: slab staging + the next tile's first K-step requested under the epilogue.  Built and measured in round 5 (same box,
// interleaved, profiles/r05_gemm_shapes.txt): 3 % SLOWER on the plain FFN launch and equal on the GELU launches in its
// first form, 20 % slower with the slab's reads hoisted above the next park - the epilogue of a 256 x 256 tile is bound
// by the 128-256 KB every CU stores at the same moment (12.5 B/clk per CU = the chip's HBM write rate), not by its LDS
// staging or by the prologue the early request hides.  The default stays the classic staging; -DCA_X_SLAB=1 builds it.
#define CA_X_SLAB 0
#endif

// grp.count > 1: a grouped launch of up to X_GROUP_MAX independent problems of the same operand form
// (ca_gemm_bf16_group): block ranges map to problems, each with plain row-major tile numbering.
#define X_GROUP_MAX 8
struct CaGemmGroup {
  CaGemmDesc d[X_GROUP_MAX];
  int first[X_GROUP_MAX];  // first tile of problem i in the group's tile list (first[0] = 0)
  int count;     // number of problems (0 = plain launch of d[0])
  int total;     // tiles in the group
  // Persistent form (vgrid > gridDim.x): the launch has one workgroup per CU; the workgroups of XCD x (blockIdx & 7)
  // take the virtual blocks x, x + 8, x + 16, ... - the first one each statically, the following ones in the order the
  // workgroups become free, through one counter per XCD (cnt[x], zero between launches: the last pull resets it).
  int vgrid;      // blocks of the virtual grid (= gridDim.x in the one-tile-per-workgroup form)
  unsigned* cnt;  // 8 counters, or null
  // Persistent form beside another resident kernel (an RCCL ring kernel during the backward of an N > 1 run): some of the
  // launch's workgroups only start when others have EXITED - i.e. after every dynamic block is taken - and a static
  // first block would then cost one whole tile time at the end of the launch.  dyn_first: every block, the first
  // included, comes from the counter; a workgroup that finds it exhausted exits at once (ca_gemm_set_compute_cus).
  int dyn_first;
};
// counter slots for persistent launches: launches that may be in flight together (two streams) use different slots
#define X_CNT_SLOTS 64
__device__ unsigned g_x_cnt[X_CNT_SLOTS][8];
#ifdef X_STAMPS
__device__ long long g_x_stamps[512 * 8 * 8];
extern "C" int ca_gemm_x_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x_stamps), (size_t)n * sizeof(long long));
}
#endif
template <int AL, int BL, bool KS>
__device__ __forceinline__ void gemm_x_body(const CaGemmGroup& grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, 128 (M) x 64 (N) each
  // ---- persistent tile loop ------------------------------------------------------------------------------------
  // One tile per workgroup when the launch covers the virtual grid (vgrid == gridDim.x).  Otherwise the workgroup
  // keeps pulling virtual blocks of its XCD until they run out: a CU that finishes early starts its next tile at once
  // (no round structure: the tiles of a "round" stop finishing together, so their output bursts spread out in time
  // and the next tile's first operand fetch runs under the previous tile's store drain).  Which workgroup computes a
  // tile never changes its result.
  const int vgrid = grp.vgrid;
  const int xcd = (int)blockIdx.x & 7;
  const int npx = ((int)gridDim.x - xcd + 7) >> 3;  // workgroups of this XCD in the launch
  const int nvx = (vgrid - xcd + 7) >> 3;           // virtual blocks of this XCD
  const int ndyn = nvx > npx ? nvx - npx : 0;       // ... handed out dynamically
  // LDS: A0 | A1 | B0 | B1 (the two operand stages, 128 KiB), then one 4-KiB epilogue slab per wave (slab_park).
  // The word pair that carries the next block's index sits at the head of wave 0's slab: thread 0 writes it at the
  // start of a tile (its own wave's previous epilogue is over by program order), every wave reads it behind the main
  // loop - at least one K-step barrier after the write - and in FRONT of the barrier that ends the tile's LDS reads;
  // only behind that barrier does wave 0's epilogue overwrite it.
  constexpr bool XSLAB = CA_X_SLAB != 0;
  volatile unsigned* nextw = (volatile unsigned*)(smem + (XSLAB ? X_SLAB_BASE : X_LDS_BYTES));  // two words, alternating per iteration
  char* const slab = smem + X_SLAB_BASE + wave * SLAB_BYTES;
  // Loader state lives outside the tile loop: with one problem per launch (the same descriptor for every tile) the
  // NEXT tile's loaders are set up and its first K-step is requested behind the barrier that ends this tile's LDS
  // reads, i.e. under this tile's epilogue (`pre`) - the operand stages are free by then, the epilogue only touches
  // the slabs.  K-major operands and plain MN-major ones stream through a scalar base (KMajorStream / MNMajorStream);
  // MN-major operands with segmented rows (KS) keep the per-lane pointer walk.
  KMajorStream<4> la_k, lb_k;
  MNMajorStream<4, 32> la_f, lb_f;
  MNMajorLoader<4, 32, KS> la_m, lb_m;
  bool pre = false;
  int pre_tm = 0, pre_tn = 0;
  int vb = (int)blockIdx.x;
  int dbase = npx, dcount = ndyn;  // dynamic blocks of this XCD: local indices dbase .. dbase + dcount - 1
  if (grp.cnt != nullptr && grp.dyn_first) {
    dbase = 0;
    dcount = nvx;
    if (tid == 0) {
      const unsigned i = __hip_atomic_fetch_add(&grp.cnt[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (i == (unsigned)(dcount + npx - 1)) __hip_atomic_store(&grp.cnt[xcd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nextw[1] = i;  // (the word of odd iterations: rewritten at the start of iteration 1, K-step barriers after this read)
    }
    __syncthreads();
    const unsigned i = (unsigned)__builtin_amdgcn_readfirstlane((int)nextw[1]);
    vb = i < (unsigned)dcount ? xcd + 8 * (int)i : vgrid;
  }
  for (int iter = 0; vb < vgrid; ++iter) {
  int vb_next = vgrid;
  if (grp.cnt != nullptr && tid == 0) {
    // this workgroup's NEXT block (the answer is read after the tile): relaxed device-scope counter
    const unsigned i = __hip_atomic_fetch_add(&grp.cnt[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i == (unsigned)(dcount + npx - 1)) __hip_atomic_store(&grp.cnt[xcd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nextw[iter & 1] = i;
  }
  bool live = true;
  int which = 0, gt = 0;
  if (!pre && grp.count > 1) {
    // grouped launch: the group's tiles form one list (problem after problem, each in run order) and XCD x
    // (= block & 7) takes the x-th run of ceil(total / 8) of them: every XCD gets the same number of tiles
    // whatever the shapes, and a run covers a compact band of one or two problems.
    gt = (vb & 7) * (vgrid >> 3) + (vb >> 3);
    live = gt < grp.total;
    if (live) {
#pragma unroll
      for (int i = 1; i < X_GROUP_MAX; ++i)
        if (grp.count > i && gt >= grp.first[i]) which = i;
    }
  }
  const CaGemmDesc d = grp.d[which];
  int tm = pre_tm, tn = pre_tn;
  if (live && !pre) {
    if (grp.count > 1)
      run_tile(gt - grp.first[which], (d.M + XBM - 1) / XBM, (d.N + XBN - 1) / XBN, tm, tn);
    else
      live = tile_of_block_g<4, 8>(vb, vgrid, (d.M + XBM - 1) / XBM, (d.N + XBN - 1) / XBN, tm, tn, d.xcd_balanced != 0);
  }
  if (live) {
  const int m0 = tm * XBM, n0 = tn * XBN;
  const int z = blockIdx.z;
  const int z1 = z / d.batch2, z2 = z % d.batch2;
  const __bf16* A = (const __bf16*)d.A + z1 * d.sA1 + z2 * d.sA2;
  const __bf16* B = (const __bf16*)d.B + z1 * d.sB1 + z2 * d.sB2;

  auto init_loaders = [&](int row0, int col0) {
    if (AL == CA_KMAJOR)
      la_k.init(A, d.lda, row0, d.M, wave, lane);
    else if (KS)
      la_m.init(A, d.lda, d.a_kseg, d.a_kseg_stride, row0, d.M, wave, lane);
    else
      la_f.init(A, d.lda, row0, d.M, wave, lane);
    if (BL == CA_KMAJOR)
      lb_k.init(B, d.ldb, col0, d.N, wave, lane);
    else if (KS)
      lb_m.init(B, d.ldb, d.b_kseg, d.b_kseg_stride, col0, d.N, wave, lane);
    else
      lb_f.init(B, d.ldb, col0, d.N, wave, lane);
  };
  if (!pre) init_loaders(m0, n0);

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float xbias[8];  // the lane's bias values, requested a whole main loop ahead of the epilogue (the segmented-K
                   // instantiations - convolution weight gradients, no bias - have no registers to spare for it)
  if constexpr (!KS) epi_load_bias(d, lane, n0 + wn * 64, z1, z2, xbias);
#ifdef X_STAMPS
  long long stamp_dma = 0, stamp_bar = 0;
#endif
  const int cs_parts = ((d.N + XBN - 1) / XBN) < 8 ? ((d.N + XBN - 1) / XBN) : 8;  // tile columns sharing the column sums
  const bool do_colsum = d.a_colsum != nullptr && tn < cs_parts;
  float csum0 = 0.f, csum1 = 0.f;  // this lane's share of sum_k A[k, m] for m = m0 + wm*128 + (ih*4 + wn)*16 + (lane & 15)

  const int K = d.K;
  const int nk = (K + BK - 1) / BK;

  // Software pipeline over K-steps of 64; LDS holds two stages of each operand tile: A0 | A1 | B0 | B1 (32 KiB each).
  // One barrier per K-step, placed in FRONT of the last of the four MFMA blocks (k-half s x m-half ih, 16 MFMAs
  // each): by then every wave has read all of tile kt and tile kt+1 has landed, so the first fragments of tile kt+1
  // are read behind the barrier and their LDS latency hides under block 3.  Fragments are double-buffered in
  // registers: the reads for block b+1 are issued BETWEEN the MFMAs of block b (one read after every second MFMA,
  // all in the first half of the block) from per-lane base addresses computed once, with the stage / k-half /
  // fragment selected by the instruction's immediate offset - no address arithmetic and no MFMA-free phase between
  // blocks.  A tile's 8 LDS-DMA go out as one burst per wave into the stage the barrier just freed: waves 0-3
  // behind the barrier, waves 4-7 one block later (SIMD partners never issue their bursts together;
  // profiles/r01_gemm_ablation.txt).
  auto burst = [&](int kt) {  // this wave's share of tile kt (called once per tile, in K order)
    if (kt >= nk) return;
    char* na = smem + (kt & 1) * XTILE;
    char* nb = na + 2 * XTILE;
    const bool full = (kt + 1) * BK <= K;  // wave-uniform
    if (full) {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        if (AL == CA_MNMAJOR && KS)
          la_m.issue_one(na, wave, lane, kt * BK, K, part);
        else if (AL == CA_KMAJOR)
          la_k.issue_one(na, wave, part);
        else
          la_f.issue_one(na, wave, part);
        if (BL == CA_MNMAJOR && KS)
          lb_m.issue_one(nb, wave, lane, kt * BK, K, part);
        else if (BL == CA_KMAJOR)
          lb_k.issue_one(nb, wave, part);
        else
          lb_f.issue_one(nb, wave, part);
      }
    } else {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        if (AL == CA_MNMAJOR && KS)
          la_m.issue_one(na, wave, lane, kt * BK, K, part);
        else if (AL == CA_KMAJOR)
          la_k.issue_one_tail(na, wave, K - kt * BK, part);
        else
          la_f.issue_one_tail(na, wave, lane, K - kt * BK, part);
        if (BL == CA_MNMAJOR && KS)
          lb_m.issue_one(nb, wave, lane, kt * BK, K, part);
        else if (BL == CA_KMAJOR)
          lb_k.issue_one_tail(nb, wave, K - kt * BK, part);
        else
          lb_f.issue_one_tail(nb, wave, lane, K - kt * BK, part);
      }
    }
    if (AL == CA_KMAJOR) la_k.advance();
    if (AL == CA_MNMAJOR && !KS) la_f.advance();
    if (BL == CA_KMAJOR) lb_k.advance();
    if (BL == CA_MNMAJOR && !KS) lb_f.advance();
  };
  // the partial last K-step (K % 64 != 0): zero what this wave loaded beyond K, once its LDS-DMA has landed
  auto zero_tail = [&](int kt) {
    if (kt != nk - 1 || nk * BK == K) return;
    char* na = smem + (kt & 1) * XTILE;
    char* nb = na + 2 * XTILE;
    const int krem = K - kt * BK;
    if (AL == CA_KMAJOR) la_k.zero_fix(na, wave, lane, krem);
    if (AL == CA_MNMAJOR && !KS) la_f.zero_fix(na, wave, lane, krem);
    if (BL == CA_KMAJOR) lb_k.zero_fix(nb, wave, lane, krem);
    if (BL == CA_MNMAJOR && !KS) lb_f.zero_fix(nb, wave, lane, krem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // per-lane fragment base addresses (stage 0, k-half 0, fragment 0 unless noted)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  uint32_t abase[8], bbase[4];  // K-major: [0], [1] = k-half 0, 1; MN-major: one per fragment (XOR swizzle)
  if (AL == CA_KMAJOR) {
    const int r = wm * 128 + (lane & 15);
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) abase[sh] = lds0 + r * 128 + (((4 * sh + (lane >> 4)) ^ ((r >> 1) & 7)) * 16);
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int swz = q | ((g & 1) << 2);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int c = ((((wm * 128 + f * 16) >> 3) + (pp >> 1)) ^ (swz << 1));
      abase[f] = lds0 + (8 * g + q) * 512 + c * 16 + (pp & 1) * 8;
    }
  }
  if (BL == CA_KMAJOR) {
    const int r = wn * 64 + (lane & 15);
#pragma unroll
    for (int sh = 0; sh < 2; ++sh)
      bbase[sh] = lds0 + 2 * XTILE + r * 128 + (((4 * sh + (lane >> 4)) ^ ((r >> 1) & 7)) * 16);
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int swz = q | ((g & 1) << 2);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int c = ((((wn * 64 + f * 16) >> 3) + (pp >> 1)) ^ (swz << 1));
      bbase[f] = lds0 + 2 * XTILE + (8 * g + q) * 512 + c * 16 + (pp & 1) * 8;
    }
  }
  bf16x8_t A0[4], A1[4], B0[4], B1[4];
  // fragment f (0..7 for A, 0..3 for B) of k-half SH from stage ST, all compile-time
#define X_RD_A(ST, SH, F, dst)                                                            \
  do {                                                                                    \
    if (AL == CA_KMAJOR)                                                                  \
      dst = lds_read_b128<(ST) * XTILE + (F) * 2048>(abase[SH]);                          \
    else                                                                                  \
      dst = lds_read_tr<(ST) * XTILE + (SH) * 16384, 2048>(abase[F]);                     \
  } while (0)
#define X_RD_B(ST, SH, F, dst)                                                            \
  do {                                                                                    \
    if (BL == CA_KMAJOR)                                                                  \
      dst = lds_read_b128<(ST) * XTILE + (F) * 2048>(bbase[SH]);                          \
    else                                                                                  \
      dst = lds_read_tr<(ST) * XTILE + (SH) * 16384, 2048>(bbase[F]);                     \
  } while (0)
#define X_SB __builtin_amdgcn_sched_barrier(0)
  // two MFMAs of block (m-half IH, fragments AF x BF): A fragment I against B fragments J and J+1
#define X_MM2(IH, AF, BF, I, J)                                                                                   \
  do {                                                                                                            \
    acc[(IH) * 4 + (I)][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[J], AF[I], acc[(IH) * 4 + (I)][J], 0, 0, 0); \
    acc[(IH) * 4 + (I)][(J) + 1] =                                                                                \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[(J) + 1], AF[I], acc[(IH) * 4 + (I)][(J) + 1], 0, 0, 0);       \
    X_SB;                                                                                                         \
  } while (0)
  // bias gradient from the streaming A tile (a_colsum): the K-steps are dealt round-robin to the first
  // cs_parts tile columns; inside a workgroup the four waves of an m-half share the work, wave wn taking
  // fragment i == wn of every block
  auto colsum_acc = [&](bool cs_step, int ih, bf16x8_t (&af)[4]) {
    if (AL == CA_MNMAJOR && cs_step) {
      // (wn is wave-uniform: a branch per case keeps the fragments in registers - indexing af[] by a run-time
      // value would send the whole array through scratch memory)
      f32x4_t z;
      if (wn == 0)
        z = __builtin_bit_cast(f32x4_t, af[0]);
      else if (wn == 1)
        z = __builtin_bit_cast(f32x4_t, af[1]);
      else if (wn == 2)
        z = __builtin_bit_cast(f32x4_t, af[2]);
      else
        z = __builtin_bit_cast(f32x4_t, af[3]);
      // v_dot2c_f32_bf16 with a vector of ones: two bf16 values added to the fp32 sum per instruction
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
      const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
      float s = ih == 0 ? csum0 : csum1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // through an integer word on purpose: bit-casting element q of the float vector straight to a bf16 pair
        // makes hipcc 7.2 read element 0 four times (seen in the ISA)
        const unsigned int wq = __float_as_uint(z[q]);
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, wq), ones, s, false);
      }
      if (ih == 0)
        csum0 = s;
      else
        csum1 = s;
    }
  };
  // one K-step on stage ST (compile-time); reads of the next tile come from stage 1 - ST
  auto kstep = [&](auto st_c, int kt) {
    constexpr int ST = decltype(st_c)::value;
    const bool cs_step = do_colsum && (kt % cs_parts) == tn;  // this K-step belongs to this tile column
    // waves 4-7 issue their share of tile kt+1 here - or, in the weight-gradient form (both operands MN-major), one
    // MFMA block later, when their SIMD partners' bursts (issued behind the barrier) are over: 2 685 against 2 802
    // cycles per K-step there, but 2 840 against 2 590 for the K-major forms (s_memtime stamps, tools/archive/dev_x_stamps.py)
    constexpr bool LATE_BURST = AL == CA_MNMAJOR && BL == CA_MNMAJOR && !KS;
    if (!LATE_BURST && wave >= 4) burst(kt + 1);
    // block 0: A(s0, m-half 0) x B(s0); reads A(s0, m-half 1)
    lds_wait(B0);
    lds_wait(A0);
    colsum_acc(cs_step, 0, A0);
    X_SB;
    __builtin_amdgcn_s_setprio(1);
    X_MM2(0, A0, B0, 0, 0); X_RD_A(ST, 0, 4, A1[0]); X_SB;
    X_MM2(0, A0, B0, 0, 2); X_RD_A(ST, 0, 5, A1[1]); X_SB;
    X_MM2(0, A0, B0, 1, 0); X_RD_A(ST, 0, 6, A1[2]); X_SB;
    X_MM2(0, A0, B0, 1, 2); X_RD_A(ST, 0, 7, A1[3]); X_SB;
    X_MM2(0, A0, B0, 2, 0);
    X_MM2(0, A0, B0, 2, 2);
    X_MM2(0, A0, B0, 3, 0);
    X_MM2(0, A0, B0, 3, 2);
    __builtin_amdgcn_s_setprio(0);
    if (LATE_BURST && wave >= 4) burst(kt + 1);
    // block 1: A(s0, m-half 1) x B(s0); reads B(s1) and A(s1, m-half 0)
    lds_wait(A1);
    colsum_acc(cs_step, 1, A1);
    X_SB;
    __builtin_amdgcn_s_setprio(1);
    X_MM2(1, A1, B0, 0, 0); X_RD_B(ST, 1, 0, B1[0]); X_SB;
    X_MM2(1, A1, B0, 0, 2); X_RD_B(ST, 1, 1, B1[1]); X_SB;
    X_MM2(1, A1, B0, 1, 0); X_RD_B(ST, 1, 2, B1[2]); X_SB;
    X_MM2(1, A1, B0, 1, 2); X_RD_B(ST, 1, 3, B1[3]); X_SB;
    X_MM2(1, A1, B0, 2, 0); X_RD_A(ST, 1, 0, A0[0]); X_SB;
    X_MM2(1, A1, B0, 2, 2); X_RD_A(ST, 1, 1, A0[1]); X_SB;
    X_MM2(1, A1, B0, 3, 0); X_RD_A(ST, 1, 2, A0[2]); X_SB;
    X_MM2(1, A1, B0, 3, 2); X_RD_A(ST, 1, 3, A0[3]); X_SB;
    __builtin_amdgcn_s_setprio(0);
    // block 2: A(s1, m-half 0) x B(s1); reads A(s1, m-half 1)
    lds_wait(B1);
    lds_wait(A0);
    colsum_acc(cs_step, 0, A0);
    X_SB;
    __builtin_amdgcn_s_setprio(1);
    X_MM2(0, A0, B1, 0, 0); X_RD_A(ST, 1, 4, A1[0]); X_SB;
    X_MM2(0, A0, B1, 0, 2); X_RD_A(ST, 1, 5, A1[1]); X_SB;
    X_MM2(0, A0, B1, 1, 0); X_RD_A(ST, 1, 6, A1[2]); X_SB;
    X_MM2(0, A0, B1, 1, 2); X_RD_A(ST, 1, 7, A1[3]); X_SB;
    X_MM2(0, A0, B1, 2, 0);
    X_MM2(0, A0, B1, 2, 2);
    X_MM2(0, A0, B1, 3, 0);
    X_MM2(0, A0, B1, 3, 2);
    __builtin_amdgcn_s_setprio(0);
    lds_wait(A1);  // the last fragment reads of tile kt have returned
    colsum_acc(cs_step, 1, A1);
    // tile kt+1 has landed (this wave's share; the barrier covers the others) and stage ST is free
#ifdef X_STAMPS
    const long long st0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef X_STAMPS
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif
    zero_tail(kt + 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef X_STAMPS
    const long long st2 = __builtin_amdgcn_s_memtime();
    stamp_dma += st1 - st0;
    stamp_bar += st2 - st1;
#endif
    if (wave < 4) burst(kt + 2);
    X_SB;
    // block 3: A(s1, m-half 1) x B(s1); reads B(s0), A(s0, m-half 0) of tile kt+1 (harmless stale data after the
    // last tile)
    __builtin_amdgcn_s_setprio(1);
    X_MM2(1, A1, B1, 0, 0); X_RD_B(1 - ST, 0, 0, B0[0]); X_SB;
    X_MM2(1, A1, B1, 0, 2); X_RD_B(1 - ST, 0, 1, B0[1]); X_SB;
    X_MM2(1, A1, B1, 1, 0); X_RD_B(1 - ST, 0, 2, B0[2]); X_SB;
    X_MM2(1, A1, B1, 1, 2); X_RD_B(1 - ST, 0, 3, B0[3]); X_SB;
    X_MM2(1, A1, B1, 2, 0); X_RD_A(1 - ST, 0, 0, A0[0]); X_SB;
    X_MM2(1, A1, B1, 2, 2); X_RD_A(1 - ST, 0, 1, A0[1]); X_SB;
    X_MM2(1, A1, B1, 3, 0); X_RD_A(1 - ST, 0, 2, A0[2]); X_SB;
    X_MM2(1, A1, B1, 3, 2); X_RD_A(1 - ST, 0, 3, A0[3]); X_SB;
    __builtin_amdgcn_s_setprio(0);
  };
#ifdef X_STAMPS
  const long long stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
  if (!pre) burst(0);  // (else: requested under the previous tile's epilogue)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  zero_tail(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef X_STAMPS
  const long long stamp_t1 = __builtin_amdgcn_s_memtime();
  const long long stamp_r1 = __builtin_amdgcn_s_memrealtime();  // 100 MHz: the loop's shader clock = d memtime / d realtime x 100 MHz
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  X_RD_B(0, 0, 0, B0[0]); X_RD_B(0, 0, 1, B0[1]); X_RD_B(0, 0, 2, B0[2]); X_RD_B(0, 0, 3, B0[3]);
  X_RD_A(0, 0, 0, A0[0]); X_RD_A(0, 0, 1, A0[1]); X_RD_A(0, 0, 2, A0[2]); X_RD_A(0, 0, 3, A0[3]);
  if (wave < 4) burst(1);
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
  }
#undef X_RD_A
#undef X_RD_B
#undef X_MM2
#undef X_SB
#ifdef X_STAMPS
  const long long stamp_t2 = __builtin_amdgcn_s_memtime();
  const long long stamp_r2 = __builtin_amdgcn_s_memrealtime();
#endif
  if (AL == CA_MNMAJOR && do_colsum) {
    // a lane's fragment holds k = 8g..8g+7 of a 32-k step: add the four lane groups, then lanes 0-15 own 16 rows
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
      float s = ih == 0 ? csum0 : csum1;
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int m = m0 + wm * 128 + (ih * 4 + wn) * 16 + (lane & 15);
      if (lane < 16 && m < d.M) d.a_colsum[(int64_t)tn * d.a_colsum_ld + m] = s;  // this tile column's share
    }
  }
  // this workgroup's next block (thread 0 wrote the word at the start of the tile, K-step barriers ago; it is read in
  // front of the barrier below, behind which wave 0's epilogue may overwrite it)
  unsigned nxt = 0;
  if (grp.cnt != nullptr) nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)nextw[iter & 1]);  // wave-uniform by construction
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave has read the last K-step: the operand stages are free
  asm volatile("" ::: "memory");
  pre = false;
  if (grp.cnt != nullptr) {
    vb_next = nxt < (unsigned)dcount ? xcd + 8 * (dbase + (int)nxt) : vgrid;
    if constexpr (!KS && XSLAB) {
      if (grp.count <= 1 && vb_next < vgrid) {
        // one problem per launch: the next tile's first K-step goes out now and lands under this tile's epilogue
        int tmn = 0, tnn = 0;
        if (tile_of_block_g<4, 8>(vb_next, vgrid, (d.M + XBM - 1) / XBM, (d.N + XBN - 1) / XBN, tmn, tnn, d.xcd_balanced != 0)) {
          init_loaders(tmn * XBM, tnn * XBN);
          burst(0);
          pre = true;
          pre_tm = tmn;
          pre_tn = tnn;
        }
      }
    }
  }
  // two 64-row halves through the shared epilogue, 16 rows at a time through the wave's slab
#pragma unroll
  for (int ih = 0; ih < 2; ++ih) {
    f32x4_t half[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) half[i][j] = acc[ih * 4 + i][j];
    if constexpr (XSLAB) {
      gemm_epilogue<false, true>(d, half, slab, wave, lane, m0 + wm * 128 + ih * 64, n0 + wn * 64, z, z1, z2, KS ? nullptr : &xbias);
    } else {
      gemm_epilogue(d, half, smem, wave, lane, m0 + wm * 128 + ih * 64, n0 + wn * 64, z, z1, z2, KS ? nullptr : &xbias);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done before it is overwritten
    }
  }
  if constexpr (!XSLAB) {
    if (grp.cnt != nullptr) __syncthreads();  // every wave is done with the staging area before the next tile lands in it
  }
#ifdef X_STAMPS
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long stamp_t3 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && iter == 0) {
      long long* o = g_x_stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
      o[0] = stamp_t1 - stamp_t0;  // prologue: first tile
      o[1] = stamp_t2 - stamp_t1;  // main loop
      o[2] = stamp_t3 - stamp_t2;  // epilogue incl. store drain
      o[3] = stamp_dma;
      o[4] = stamp_bar;
      o[5] = nk;
      o[6] = stamp_r2 - stamp_r1;  // main loop in 100-MHz ticks
    }
  }
#endif
  }  // live
  else if (grp.cnt != nullptr) {
    // a padding block of the virtual grid: no K-step barrier separates thread 0's write from the reads
    pre = false;
    __syncthreads();
    const unsigned i = (unsigned)__builtin_amdgcn_readfirstlane((int)nextw[iter & 1]);
    vb_next = i < (unsigned)dcount ? xcd + 8 * (dbase + (int)i) : vgrid;
  }
  if (grp.cnt == nullptr) break;
  vb = vb_next;
  }  // persistent tile loop
}

template <int AL, int BL, bool KS>
__global__ __launch_bounds__(512) void ca_gemm_kernel_x(const CaGemmGroup grp) {
  gemm_x_body<AL, BL, KS>(grp);
}
// The NT form (every forward GEMM of the training step) is held to 224 registers per lane: the background AdamW of the
// overlapped optimiser (misc.hip adamw_kernel, 60 -> 64 registers, one workgroup per CU) then fits beside the two waves
// per SIMD of a persistent workgroup (2 x 224 + 64 = 512).  At 225 the allocation granule of 8 makes it 2 x 232: the two
// kernels stop sharing CUs, alternate instead, and the XLS-R-2B step went from 72 to 91 ms (NOTEBOOK.md R5.6).
// (amdgpu_num_vgpr counts one half of gfx950's unified VGPR|AGPR file: 112 = 224 registers; tests/test_build.py checks
// the code object.)
template <>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(112)))
void ca_gemm_kernel_x<CA_KMAJOR, CA_KMAJOR, false>(const CaGemmGroup grp) {
  gemm_x_body<CA_KMAJOR, CA_KMAJOR, false>(grp);
}

// ---- kernel L: 256x128 tile, 8 waves (4x2, 64x64 each), 3 LDS stages, one workgroup per CU ---------------------
// For the N = d shapes of the path (out-projection, FFN2 and the data gradients that produce [tokens, d]): 128
// tiles of 256x256 leave half the chip idle, and the 128x128 kernel with its two workgroups per CU runs at the
// per-CU LDS-fill limit (2 x 32 KiB per 64-k step for 2 x 128 x 128 outputs = ~95 GB/s per CU at its in-step rate;
// MI355X_MICROARCH: 66-73 GB/s L2-served).  256x128 tiles give every CU one workgroup (240 tiles at M = 3992,
// N = 1920) at 3/4 of S's fill bytes per FLOP.  Built like kernel X: scalar-base LDS-DMA streams, fragments
// double-buffered in registers with the reads issued BETWEEN the MFMAs of the running block (immediate-offset
// addressing), one barrier per K-step in front of its last block, staggered DMA bursts (waves 0-3 right behind the
// barrier, waves 4-7 one block later).  A K-step has two blocks of 16 MFMAs (k-half 0, k-half 1) per wave.  The 48-KiB
// stages leave room for a ring of three: the LDS-DMA runs TWO tiles ahead behind a counted vmcnt (6 pieces per wave and
// tile), which is what keeps the loop fed when the optimiser's traffic beside the forward stretches the fetch latency.
#define LBM 256
#define LBN 128
#define LA_BYTES (LBM * BK * 2)  // 32 KiB
#define LB_BYTES (LBN * BK * 2)  // 16 KiB
#define L_NST 3
#define L_LDS_BYTES (L_NST * (LA_BYTES + LB_BYTES))  // 147456 >= 8 waves * 64*68*4 (139264) epilogue staging

template <int AL, int BL>
__global__ __launch_bounds__(512) void ca_gemm_kernel_l(const CaGemmDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves, 64x64 each
  int tm, tn;
  if (!tile_of_block<4, 8>(blockIdx.x, (d.M + LBM - 1) / LBM, (d.N + LBN - 1) / LBN, tm, tn, d.xcd_balanced != 0)) return;
  const int m0 = tm * LBM, n0 = tn * LBN;
  const int z = blockIdx.z;
  const int z1 = z / d.batch2, z2 = z % d.batch2;
  const __bf16* A = (const __bf16*)d.A + z1 * d.sA1 + z2 * d.sA2;
  const __bf16* B = (const __bf16*)d.B + z1 * d.sB1 + z2 * d.sB2;
  const int K = d.K;
  const int nk = (K + BK - 1) / BK;

  KMajorStream<4> la_k;
  KMajorStream<2> lb_k;
  MNMajorStream<4, 32> la_f;
  MNMajorStream<2, 16> lb_f;
  if (AL == CA_KMAJOR)
    la_k.init(A, d.lda, m0, d.M, wave, lane);
  else
    la_f.init(A, d.lda, m0, d.M, wave, lane);
  if (BL == CA_KMAJOR)
    lb_k.init(B, d.ldb, n0, d.N, wave, lane);
  else
    lb_f.init(B, d.ldb, n0, d.N, wave, lane);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float lbias[8];  // the lane's bias values, requested a whole main loop ahead of the epilogue
  epi_load_bias(d, lane, n0 + wn * 64, z1, z2, lbias);

  // LDS: A0 | A1 | A2 (32 KiB each) | B0 | B1 | B2 (16 KiB each)
  int bst = 0;  // stage of this wave's next burst (every wave issues its share of every tile, in order)
  auto burst = [&](int kt) {  // this wave's share of tile kt: 4 A pieces + 2 B pieces
    if (kt >= nk) return;
    char* na = smem + bst * LA_BYTES;
    char* nb = smem + L_NST * LA_BYTES + bst * LB_BYTES;
    bst = bst == L_NST - 1 ? 0 : bst + 1;
    const bool full = (kt + 1) * BK <= K;  // wave-uniform
    if (full) {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        if (AL == CA_KMAJOR)
          la_k.issue_one(na, wave, part);
        else
          la_f.issue_one(na, wave, part);
        if (part < 2) {
          if (BL == CA_KMAJOR)
            lb_k.issue_one(nb, wave, part);
          else
            lb_f.issue_one(nb, wave, part);
        }
      }
    } else {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        if (AL == CA_KMAJOR)
          la_k.issue_one_tail(na, wave, K - kt * BK, part);
        else
          la_f.issue_one_tail(na, wave, lane, K - kt * BK, part);
        if (part < 2) {
          if (BL == CA_KMAJOR)
            lb_k.issue_one_tail(nb, wave, K - kt * BK, part);
          else
            lb_f.issue_one_tail(nb, wave, lane, K - kt * BK, part);
        }
      }
    }
    if (AL == CA_KMAJOR) la_k.advance(); else la_f.advance();
    if (BL == CA_KMAJOR) lb_k.advance(); else lb_f.advance();
  };
  auto zero_tail = [&](int kt) {
    if (kt != nk - 1 || nk * BK == K) return;
    char* na = smem + (kt % L_NST) * LA_BYTES;
    char* nb = smem + L_NST * LA_BYTES + (kt % L_NST) * LB_BYTES;
    const int krem = K - kt * BK;
    if (AL == CA_KMAJOR) la_k.zero_fix(na, wave, lane, krem); else la_f.zero_fix(na, wave, lane, krem);
    if (BL == CA_KMAJOR) lb_k.zero_fix(nb, wave, lane, krem); else lb_f.zero_fix(nb, wave, lane, krem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // per-lane fragment base addresses (stage 0; K-major: one per k-half, fragment = immediate offset;
  // MN-major: one per fragment, k-half = immediate offset)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  uint32_t abase[4], bbase[4];
  if (AL == CA_KMAJOR) {
    const int r = wm * 64 + (lane & 15);
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) abase[sh] = lds0 + r * 128 + (((4 * sh + (lane >> 4)) ^ ((r >> 1) & 7)) * 16);
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int swz = q | ((g & 1) << 2);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int c = ((((wm * 64 + f * 16) >> 3) + (pp >> 1)) ^ (swz << 1));
      abase[f] = lds0 + (8 * g + q) * 512 + c * 16 + (pp & 1) * 8;
    }
  }
  if (BL == CA_KMAJOR) {
    const int r = wn * 64 + (lane & 15);
#pragma unroll
    for (int sh = 0; sh < 2; ++sh)
      bbase[sh] = lds0 + L_NST * LA_BYTES + r * 128 + (((4 * sh + (lane >> 4)) ^ ((r >> 1) & 7)) * 16);
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int swz = q | ((g & 1) << 2);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int c = ((((wn * 64 + f * 16) >> 3) + (pp >> 1)) ^ (swz << 1));
      bbase[f] = lds0 + L_NST * LA_BYTES + (8 * g + q) * 256 + c * 16 + (pp & 1) * 8;
    }
  }
  // (DS immediate offsets end at 64 KiB: the third A stage gets base registers of its own)
  uint32_t abase2[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) abase2[f] = abase[f] + 2 * LA_BYTES;
  bf16x8_t A0[4], A1[4], B0[4], B1[4];
#define L_RD_A(ST, SH, F, dst)                                                                      \
  do {                                                                                              \
    if (AL == CA_KMAJOR)                                                                            \
      dst = lds_read_b128<((ST) % 2) * LA_BYTES + (F) * 2048>((ST) == 2 ? abase2[SH] : abase[SH]);  \
    else                                                                                            \
      dst = lds_read_tr<((ST) % 2) * LA_BYTES + (SH) * 16384, 2048>((ST) == 2 ? abase2[F] : abase[F]); \
  } while (0)
#define L_RD_B(ST, SH, F, dst)                                                  \
  do {                                                                          \
    if (BL == CA_KMAJOR)                                                        \
      dst = lds_read_b128<(ST) * LB_BYTES + (F) * 2048>(bbase[SH]);             \
    else                                                                        \
      dst = lds_read_tr<(ST) * LB_BYTES + (SH) * 8192, 1024>(bbase[F]);         \
  } while (0)
#define L_SB __builtin_amdgcn_sched_barrier(0)
#define L_MM2(AF, BF, I, J)                                                                             \
  do {                                                                                                  \
    acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[J], AF[I], acc[I][J], 0, 0, 0);              \
    acc[I][(J) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[(J) + 1], AF[I], acc[I][(J) + 1], 0, 0, 0); \
    L_SB;                                                                                               \
  } while (0)
  auto kstep = [&](auto st_c, int kt) {
    constexpr int ST = decltype(st_c)::value;
    constexpr int NS = (ST + 1) % L_NST;  // stage of tile kt + 1
    if (wave >= 4) burst(kt + 2);
    // block 0: k-half 0 of tile kt; reads k-half 1 (same stage)
    lds_wait(B0);
    lds_wait(A0);
    L_SB;
    __builtin_amdgcn_s_setprio(1);
    L_MM2(A0, B0, 0, 0); L_RD_B(ST, 1, 0, B1[0]); L_SB;
    L_MM2(A0, B0, 0, 2); L_RD_B(ST, 1, 1, B1[1]); L_SB;
    L_MM2(A0, B0, 1, 0); L_RD_B(ST, 1, 2, B1[2]); L_SB;
    L_MM2(A0, B0, 1, 2); L_RD_B(ST, 1, 3, B1[3]); L_SB;
    L_MM2(A0, B0, 2, 0); L_RD_A(ST, 1, 0, A1[0]); L_SB;
    L_MM2(A0, B0, 2, 2); L_RD_A(ST, 1, 1, A1[1]); L_SB;
    L_MM2(A0, B0, 3, 0); L_RD_A(ST, 1, 2, A1[2]); L_SB;
    L_MM2(A0, B0, 3, 2); L_RD_A(ST, 1, 3, A1[3]); L_SB;
    __builtin_amdgcn_s_setprio(0);
    lds_wait(B1);
    lds_wait(A1);  // the last fragment reads of tile kt have returned
    // tile kt+1 has landed (this wave's share; the barrier covers the others), tile kt+2 may still be in flight
    if (kt + 2 < nk)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    zero_tail(kt + 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave < 4) burst(kt + 3);  // into stage ST: every wave has finished reading tile kt
    L_SB;
    // block 1: k-half 1 of tile kt; reads k-half 0 of tile kt+1 (harmless stale data after the last tile)
    __builtin_amdgcn_s_setprio(1);
    L_MM2(A1, B1, 0, 0); L_RD_B(NS, 0, 0, B0[0]); L_SB;
    L_MM2(A1, B1, 0, 2); L_RD_B(NS, 0, 1, B0[1]); L_SB;
    L_MM2(A1, B1, 1, 0); L_RD_B(NS, 0, 2, B0[2]); L_SB;
    L_MM2(A1, B1, 1, 2); L_RD_B(NS, 0, 3, B0[3]); L_SB;
    L_MM2(A1, B1, 2, 0); L_RD_A(NS, 0, 0, A0[0]); L_SB;
    L_MM2(A1, B1, 2, 2); L_RD_A(NS, 0, 1, A0[1]); L_SB;
    L_MM2(A1, B1, 3, 0); L_RD_A(NS, 0, 2, A0[2]); L_SB;
    L_MM2(A1, B1, 3, 2); L_RD_A(NS, 0, 3, A0[3]); L_SB;
    __builtin_amdgcn_s_setprio(0);
  };
  burst(0);
  burst(1);
  if (nk > 1)
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  zero_tail(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  L_RD_B(0, 0, 0, B0[0]); L_RD_B(0, 0, 1, B0[1]); L_RD_B(0, 0, 2, B0[2]); L_RD_B(0, 0, 3, B0[3]);
  L_RD_A(0, 0, 0, A0[0]); L_RD_A(0, 0, 1, A0[1]); L_RD_A(0, 0, 2, A0[2]); L_RD_A(0, 0, 3, A0[3]);
  if (wave < 4) burst(2);
  for (int kt = 0; kt < nk; kt += 3) {
    kstep(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
    if (kt + 2 < nk) kstep(std::integral_constant<int, 2>{}, kt + 2);
  }
#undef L_RD_A
#undef L_RD_B
#undef L_MM2
#undef L_SB
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  gemm_epilogue(d, acc, smem, wave, lane, m0 + wm * 64, n0 + wn * 64, z, z1, z2, &lbias);
}

// ---- skinny-M kernel: M <= 32 rows (one decoded token per clip) -------------------------------------
// C[m, n] = epilogue(alpha * sum_k A[m, k] W[n, k]): a weight-streaming problem (every weight byte is used once),
// so there is no LDS staging: each wave owns 16 output columns and a quarter of K, loads its W fragment
// (16 rows x 64 B) and the matching A fragment(s) straight from global memory into MFMA operands, eight k-steps
// in flight; the four waves of a workgroup add their partial tiles through LDS and wave 0 runs the epilogue.
// N/16 workgroups: 64 (N = 1024) to 3242 (the 51865-entry vocabulary).  MB = row blocks of 16: a batch of 17..32 clips
// takes a second A fragment and accumulator against the SAME weight fragment (round 3: the launches and the weight
// bytes of a decoded token are shared by twice the clips).
// LN (CaGemmDesc.a_ln_gamma): the A rows are LayerNorm(A rows), formed by every workgroup in its prologue (the
// arithmetic of ln_fwd_kernel, chunk by chunk: a wave per row, the normalised row rounded to bf16 as the LayerNorm
// launch would have stored it) into an LDS image the MFMA operand loads then read - the LayerNorm launch in front of
// the q|k|v and fc1 projections of a decoded token (4.9 us each from dispatch to completion, a quarter of them a
// memory round trip) disappears from the per-token chain; bit-identical to the two launches.  The first eight weight
// fragments of every wave (all of them at K = 1024) are asked for BEFORE the prologue, so the statistics run under
// the weight fetch.
#define SKINNY_LN_PAD 32  // bf16 elements between LDS rows beyond K: 64 B, lanes r and r + 1 then sit 16 banks apart
// NCH: 64-lane rounds of 8-element chunks a row takes (K <= 512 NCH), 0 = no LayerNorm prologue
// Round 5: 33 .. 128 rows (an evaluation run decodes more clips per step than 32 - R/config/evaluation.yaml:20,
// R/src/coral/evaluate.py:56-60) run the SAME 32-row kernel on a two-dimensional grid: blockIdx.y picks the block of 32
// rows.  A workgroup then streams 64 KB of A per 1024 k beside its 32 KB of weights, as at 32 rows (four row blocks in one
// workgroup measured 17.7 us per launch at 64 rows against 8 at 32: the A rows, re-read by every workgroup, were 4/5 of
// its bytes); the weight fragment is read by the 2 - 4 workgroups of a column block, once from HBM and then from L2.
// Round 6: NT = output columns per workgroup (16, 8 or 4: MFMA rows NT .. 15 repeat rows 0 .. NT-1 and are dropped), U =
// k-steps a wave asks for at once.  What a launch of this kernel costs is the bytes its BUSIEST CU takes in (~10 B/clk
// per CU from HBM, the weights; ~3x that from L2, the activation rows every workgroup reads): at N = 1024 sixteen-column
// workgroups put the whole weight matrix through 64 CUs (fc2 of a decoded token at 16 clips: 128 KB of weights + 128 KB
// of activations per CU, in four dependent rounds of eight k-steps: 13 us); four-column workgroups on 256 CUs with the
// wave's whole K quarter in flight take 32 KB of weights each.  Per output element the same K split and the same MFMA
// chain: the same bits whatever NT and U.
template <int MB, int NCH = 0, int U = 8, int NT = 16>
__global__ __launch_bounds__(256) void ca_gemm_skinny_kernel(const CaGemmDesc d) {
  constexpr bool LN = NCH > 0;
  static_assert(!LN || U == 8, "the LayerNorm prologue keeps eight weight fragments in flight");
  static_assert(NT == 16 || NT == 8 || NT == 4, "4, 8 or 16 columns per workgroup");
  const int mrow0 = (int)blockIdx.y * 16 * MB;  // first row of this workgroup's row block
  constexpr int NC = LN ? NCH : 1;
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];
  float(*part)[MB * 16 * 16] = (float(*)[MB * 16 * 16]) sk_smem;
  unsigned short* const xs = (unsigned short*)(sk_smem + 4 * MB * 256 * sizeof(float));  // LN only: [16 MB][K + pad]
  const int xpitch = d.K + SKINNY_LN_PAD;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * NT;
  const int r = lane & 15, g = lane >> 4;
  const __bf16* A = (const __bf16*)d.A;
  const __bf16* W = (const __bf16*)d.B;
  const int rr = r & (NT - 1);  // (MFMA rows beyond NT: copies, same addresses - one request)
  const int nrow = n0 + rr < d.N ? n0 + rr : d.N - 1;
  const __bf16* wp = W + (int64_t)nrow * d.ldb + 8 * g;
  const __bf16* ap[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mrow0 + 16 * mb + r;
    ap[mb] = A + (int64_t)(m < d.M ? m : d.M - 1) * d.lda + 8 * g;
  }
  const int ksteps = (d.K + 31) / 32;
  const int per = (ksteps + 3) / 4;
  const int ks0 = wave * per, ks1 = (ks0 + per < ksteps) ? ks0 + per : ksteps;
  f32x4_t acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, (f32x4_t){0.f, 0.f, 0.f, 0.f});
  // The kernel is a short chain of memory round trips behind a ~4 us dependent launch (measured: 4.0 us per launch
  // in a graph at N = K = 1024, +1 us per 24 KB a workgroup streams - the per-CU fill rate - tools/archive/dev_skinny_time.py),
  // so the epilogue's operands (bias, residual, destination row) are asked for up front, beside the first K-steps.
  const int epi = d.epilogue;
  const bool wave0 = wave == 0;
  float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_res[MB][4];
  int64_t crow[MB];
  // c_split_n: columns [c_split_n, N) go to a second output (C_hi, ldc_hi; column index n - c_split_n) and only they
  // take the c_row_index rows - q and the cached K|V rows of a decoded token from one launch.  n0 is a multiple of
  // 16 and so is c_split_n: a workgroup is on one side.
  const bool hi = d.c_split_n > 0 && n0 >= d.c_split_n;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mrow0 + 16 * mb + r;
    crow[mb] = m;
#pragma unroll
    for (int e = 0; e < 4; ++e) e_res[mb][e] = 0.f;
    if (wave0 && m < d.M) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + 4 * g + e;
        if (n < d.N && 4 * g + e < NT) {
          if (mb == 0 && d.bias) e_bias[e] = d.bias[n];
          if (epi == CA_EPI_GELU_RESIDUAL || epi == CA_EPI_RESIDUAL)
            e_res[mb][e] = bf2f(((const unsigned short*)d.R)[(int64_t)m * d.ldr + n]);
        }
      }
      if (d.c_row_index && (d.c_split_n == 0 || hi)) crow[mb] = (int64_t)m * d.c_row_mul + d.c_row_index[m];
    }
  }
  void* const Cdst = hi ? d.C_hi : d.C;
  const int64_t ldcd = hi ? d.ldc_hi : d.ldc;
  const int ncol0 = hi ? d.c_split_n : 0;
  bf16x8_t wf0[8];  // LN: the first iteration's weight fragments, in flight across the prologue
  if constexpr (LN) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = (ks0 + u) * 32 + 8 * g;
      wf0[u] = (ks0 + u < ks1 && k < d.K) ? *(const bf16x8_t*)(wp + (int64_t)(ks0 + u) * 32) : zero;
    }
    // rows wave, wave + 4, ...: all of a wave's rows are asked for before the first statistic (one round trip)
    constexpr int RW = MB * 4;
    const int C = d.K, nchunk = C >> 3;
    const unsigned short* xr = (const unsigned short*)d.A;
    u16x8_t raw[RW][NC];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int row = mrow0 + wave + 4 * i;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        raw[i][c] = (row < d.M && ch < nchunk) ? *(const u16x8_t*)(xr + (int64_t)row * d.lda + ch * 8)
                                               : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    f32x4_t gq[NC][2], bq[NC][2];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        gq[c][0] = *(const f32x4_t*)(d.a_ln_gamma + ch * 8);
        gq[c][1] = *(const f32x4_t*)(d.a_ln_gamma + ch * 8 + 4);
        bq[c][0] = *(const f32x4_t*)(d.a_ln_beta + ch * 8);
        bq[c][1] = *(const f32x4_t*)(d.a_ln_beta + ch * 8 + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int lrow = wave + 4 * i;  // row of the LDS image; global row mrow0 + lrow
      const int row = mrow0 + lrow;
      if (row >= d.M) break;  // (wave-uniform)
      float sx = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (lane + c * 64 < nchunk) {
#pragma unroll
          for (int e = 0; e < 8; ++e) sx += bf2f(raw[i][c][e]);
        }
      }
      const float mean = wave_sum(sx) / (float)C;
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (lane + c * 64 < nchunk) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dlt = bf2f(raw[i][c][e]) - mean;
            s2 += dlt * dlt;
          }
        }
      }
      const float rstd = rsqrtf(wave_sum(s2) / (float)C + d.a_ln_eps);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
          u16x8_t o8;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            o8[e] = f2bf(ln_apply(bf2f(raw[i][c][e]), mean, rstd, e < 4 ? gq[c][0][e] : gq[c][1][e - 4],
                                  e < 4 ? bq[c][0][e] : bq[c][1][e - 4]));
          *(u16x8_t*)(xs + (int64_t)lrow * xpitch + ch * 8) = o8;
        }
      }
    }
    __syncthreads();
  }
  for (int ks = ks0; ks < ks1; ks += U) {
    bf16x8_t wf[U], af[MB][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (ks + u) * 32 + 8 * g;
      const bool ok = ks + u < ks1 && k < d.K;  // K is a multiple of 8 (lda/ldb rule), so a chunk is all-or-nothing
      if constexpr (LN) {
        wf[u] = ks == ks0 ? wf0[u] : (ok ? *(const bf16x8_t*)(wp + (int64_t)(ks + u) * 32) : zero);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          af[mb][u] = (ok && mrow0 + 16 * mb + r < d.M) ? *(const bf16x8_t*)(xs + (int64_t)(16 * mb + r) * xpitch + k) : zero;
      } else {
        wf[u] = ok ? *(const bf16x8_t*)(wp + (int64_t)(ks + u) * 32) : zero;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          af[mb][u] = (ok && mrow0 + 16 * mb + r < d.M) ? *(const bf16x8_t*)(ap[mb] + (int64_t)(ks + u) * 32) : zero;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[mb][u], acc[mb], 0, 0, 0);
  }
  // D[n = 4g + e][m = r]: lane holds 4 consecutive n of row m
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][mb * 256 + r * 16 + 4 * g + e] = acc[mb][e];
  __syncthreads();
  if (!wave0) return;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (mrow0 + 16 * mb + r >= d.M) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + 4 * g + e;
      if (n >= d.N || 4 * g + e >= NT) continue;
      const int i = mb * 256 + r * 16 + 4 * g + e;
      float v = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
      v = v * d.alpha + e_bias[e];
      float v2 = 0.f;
      if (epi == CA_EPI_GELU || epi == CA_EPI_GELU_RESIDUAL) {
        v2 = gelu_erf(v);
        if (epi == CA_EPI_GELU_RESIDUAL) v2 += e_res[mb][e];
      } else if (epi == CA_EPI_RESIDUAL) {
        v += e_res[mb][e];
      }
      const int64_t off = crow[mb] * ldcd + (n - ncol0);
      if (Cdst) {
        if (d.out_f32)
          ((float*)Cdst)[off] = d.accumulate ? ((float*)Cdst)[off] + v : v;
        else
          ((unsigned short*)Cdst)[off] = f2bf(d.accumulate ? bf2f(((unsigned short*)Cdst)[off]) + v : v);
      }
      if ((epi == CA_EPI_GELU || epi == CA_EPI_GELU_RESIDUAL) && d.C2) ((unsigned short*)d.C2)[off] = f2bf(v2);
    }
  }
}

// ---- optional per-launch timing (bench.py's live roofline measurement) -----------------------
// When enabled, every ca_gemm_bf16 launch is bracketed by two hipEvents on the launch stream and
// its algorithmic FLOPs (2*M*N*K*batch) are recorded per template variant (index kernel*8 +
// segmented-K*4 + a_layout*2 + b_layout; kernel 0 = S, 1 = L, 2 = X).  ca_prof_end synchronises the events and returns the totals.
#include <hip/hip_ext.h>
#include <vector>
struct ProfRec {
  hipEvent_t e0, e1;
  double flops;
  int variant;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static hipEvent_t g_prof_e0 = nullptr, g_prof_e1 = nullptr;  // events of the launch being profiled
// While profiling, the kernel is launched with start/stop events attached to the dispatch itself
// (hipExtLaunchKernelGGL): they carry the kernel's own begin/end timestamps, like rocprofv3's kernel trace,
// instead of the gaps between stream operations.
#define CA_LAUNCH(kernel, grid, block, lds, stream, ...)                                                    \
  do {                                                                                                       \
    if (g_prof_e0)                                                                                           \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_prof_e0, g_prof_e1, 0, __VA_ARGS__);         \
    else                                                                                                     \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                     \
  } while (0)

extern "C" int ca_prof_begin(void) {
  g_prof.clear();
  g_prof_on = true;
  return CA_OK;
}
extern "C" int ca_prof_end(double* ms, int64_t* count, double* flops) {
  g_prof_on = false;
  for (int v = 0; v < 24; ++v) {
    ms[v] = 0.0;
    count[v] = 0;
    flops[v] = 0.0;
  }
  for (auto& r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) {
      ca_set_error("ca_prof_end: event query failed");
      return CA_ERR_LAUNCH;
    }
    ms[r.variant] += t;
    count[r.variant] += 1;
    flops[r.variant] += r.flops;
    hipEventDestroy(r.e0);
    hipEventDestroy(r.e1);
  }
  g_prof.clear();
  return CA_OK;
}

extern "C" int ca_gemm_debug_general_epilogue(int on) {
  const int v = on ? 1 : 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ca_epi_general), &v, sizeof(int)) != hipSuccess) {
    ca_set_error("ca_gemm_debug_general_epilogue: hipMemcpyToSymbol failed");
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}
// CUs the compute kernels may count on (ca_gemm_set_compute_cus): 0 = all of them.  Set by the trainer of an N > 1 run,
// where a collective's kernel holds some CUs for most of the backward: persistent launches are sized to the rest and
// hand out every tile dynamically (CaGemmGroup.dyn_first), and the kernel-choice rule counts rounds on the rest.
static int g_compute_cus = 0;
extern "C" int ca_gemm_set_compute_cus(int n) {
  g_compute_cus = n > 0 ? n : 0;
  return CA_OK;
}
// see include/coral_amd.h: the forward kernel that holds the most registers is kernel X's K-major form
extern "C" int ca_background_update_fits(int32_t* regs) {
  hipFuncAttributes ax, aw;
  const hipError_t e1 = hipFuncGetAttributes(&ax, (const void*)ca_gemm_kernel_x<0, 0, false>);
  const hipError_t e2 = hipFuncGetAttributes(&aw, ca_adamw_background_kernel());
  if (e1 != hipSuccess || e2 != hipSuccess) {
    ca_set_error("ca_background_update_fits: hipFuncGetAttributes: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return CA_ERR_LAUNCH;
  }
  if (regs) {
    regs[0] = ax.numRegs;
    regs[1] = aw.numRegs;
  }
  const int gx = (ax.numRegs + 7) / 8 * 8, gw = (aw.numRegs + 7) / 8 * 8;
  return 2 * gx + gw <= 512 ? 1 : 0;
}
static unsigned x_device_cus() {
  static const unsigned ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return (unsigned)(n >= 8 ? (n / 8) * 8 : 8);
  }();
  return ncu;
}
static unsigned x_compute_cus() {
  const unsigned ncu = x_device_cus();
  if (g_compute_cus <= 0) return ncu;
  const unsigned c = (unsigned)(g_compute_cus >= 8 ? (g_compute_cus / 8) * 8 : 8);
  return c < ncu ? c : ncu;
}
static int g_force_kernel = 0;  // 0 auto, 1 force 128x128, 2 force 256x128 (tests / tuning)
extern "C" int ca_gemm_force_kernel(int which) {
  g_force_kernel = which;
  return CA_OK;
}
static int ca_gemm_launch(const CaGemmDesc* desc, void* stream);
static int g_last_kind = 0;  // kernel chosen by the last launch: 0 = S, 1 = L, 2 = X

extern "C" int ca_gemm_bf16(const CaGemmDesc* desc, void* stream) {
  if (!g_prof_on || desc == nullptr) return ca_gemm_launch(desc, stream);
  ProfRec r;
  hipEventCreate(&r.e0);
  hipEventCreate(&r.e1);
  r.flops = 2.0 * desc->M * (double)desc->N * desc->K * desc->batch1 * desc->batch2;
  g_prof_e0 = r.e0;
  g_prof_e1 = r.e1;
  const int rc = ca_gemm_launch(desc, stream);
  g_prof_e0 = g_prof_e1 = nullptr;
  r.variant = g_last_kind * 8 + ((desc->a_kseg > 0 || desc->b_kseg > 0) ? 4 : 0) + (desc->a_layout ? 2 : 0) +
              (desc->b_layout ? 1 : 0);
  g_prof.push_back(r);
  return rc;
}

// ---- fp8 (OCP e4m3) forward GEMM: C = epilogue(alpha * sa * sb * A B^T), A [M][K], B [N][K] one byte per element ----
// BASELINE configs[4] ("fp8 weights, CDNA4 fp8 MFMA").  v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales
// (the per-tensor scales sa, sb are device scalars folded into alpha): 4x the K of the bf16 MFMA at twice its cycles.
// A 128-byte row of the LDS image now holds 128 k, so the tile images, the loaders and the XOR swizzle are the bf16
// kernel's byte for byte; a lane's operand is the 32 consecutive bytes (two 16-B chunks) of its row and lane group,
// for A and B alike - the dot product does not care which k a lane group owns as long as both sides agree.
// Kernel S structure: 128 x 128 x 128 tile, 4 waves (64 x 64 each), 2 workgroups per CU.
typedef __attribute__((ext_vector_type(8))) int fp8x32_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__global__ __launch_bounds__(256) void ca_gemm_fp8_kernel(const CaGemmDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  if (!tile_of_block<8, 8>(blockIdx.x, (d.M + BM - 1) / BM, (d.N + BN - 1) / BN, tm, tn)) return;
  const int m0 = tm * BM, n0 = tn * BN;
  KMajorStream<4> la, lb;
  la.init_bytes((const char*)d.A, d.lda, m0, d.M, wave, lane);
  lb.init_bytes((const char*)d.B, d.ldb, n0, d.N, wave, lane);
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int K = d.K;
  constexpr int FBK = 128;  // k per K-step
  const int nk = (K + FBK - 1) / FBK;
  auto burst = [&](int kt) {
    if (kt >= nk) return;
    char* na = smem + (kt & 1) * TILE_BYTES;
    char* nb = na + 2 * TILE_BYTES;
    const bool full = (kt + 1) * FBK <= K;
    if (full) {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        la.issue_one(na, wave, part);
        lb.issue_one(nb, wave, part);
      }
    } else {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        la.issue_one_tail(na, wave, K - kt * FBK, part);
        lb.issue_one_tail(nb, wave, K - kt * FBK, part);
      }
    }
    la.advance();
    lb.advance();
  };
  auto zero_tail = [&](int kt) {
    if (kt != nk - 1 || nk * FBK == K) return;
    char* na = smem + (kt & 1) * TILE_BYTES;
    la.zero_fix(na, wave, lane, K - kt * FBK);
    lb.zero_fix(na + 2 * TILE_BYTES, wave, lane, K - kt * FBK);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const int g = lane >> 4;
  uint32_t abase[2], bbase[2];  // the two 16-B chunks of this lane's 32-byte operand
  {
    const int ra = wm * 64 + (lane & 15), rb = wn * 64 + (lane & 15);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      abase[h] = lds0 + ra * 128 + (((2 * g + h) ^ ((ra >> 1) & 7)) * 16);
      bbase[h] = lds0 + 2 * TILE_BYTES + rb * 128 + (((2 * g + h) ^ ((rb >> 1) & 7)) * 16);
    }
  }
  burst(0);
  auto kstep = [&](auto st_c, int kt) {
    constexpr int ST = decltype(st_c)::value;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tile kt landed; my reads of kt-1 done
    zero_tail(kt);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    burst(kt + 1);
    bf16x8_t al[4], ah[4], bl[4], bh[4];
#define F8_RD(dst, base, F) dst = lds_read_b128<ST * TILE_BYTES + (F) * 2048>(base)
    F8_RD(al[0], abase[0], 0); F8_RD(ah[0], abase[1], 0); F8_RD(al[1], abase[0], 1); F8_RD(ah[1], abase[1], 1);
    F8_RD(al[2], abase[0], 2); F8_RD(ah[2], abase[1], 2); F8_RD(al[3], abase[0], 3); F8_RD(ah[3], abase[1], 3);
    F8_RD(bl[0], bbase[0], 0); F8_RD(bh[0], bbase[1], 0); F8_RD(bl[1], bbase[0], 1); F8_RD(bh[1], bbase[1], 1);
    F8_RD(bl[2], bbase[0], 2); F8_RD(bh[2], bbase[1], 2); F8_RD(bl[3], bbase[0], 3); F8_RD(bh[3], bbase[1], 3);
#undef F8_RD
    lds_wait(al);
    lds_wait(ah);
    lds_wait(bl);
    lds_wait(bh);
    fp8x32_t af[4], bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const i32x4_t a0 = __builtin_bit_cast(i32x4_t, al[i]), a1 = __builtin_bit_cast(i32x4_t, ah[i]);
      const i32x4_t b0 = __builtin_bit_cast(i32x4_t, bl[i]), b1 = __builtin_bit_cast(i32x4_t, bh[i]);
      af[i] = (fp8x32_t){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      bq[i] = (fp8x32_t){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bq[j], af[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                     0x7f7f7f7f);
  };
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  CaGemmDesc dd = d;
  dd.alpha = d.alpha * (d.a_scale ? *d.a_scale : 1.f) * (d.b_scale ? *d.b_scale : 1.f);
  if (d.a_row_scale) {  // one dequantisation factor per row of A (ca_layernorm_fwd_fp8): a lane's accumulators of
                        // fragment row i all belong to output row m0 + wm*64 + 16 i + (lane & 15)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + (lane & 15);
      const float rs = d.a_row_scale[m < d.M ? m : d.M - 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= rs;
    }
  }
  gemm_epilogue(dd, acc, smem, wave, lane, m0 + wm * 64, n0 + wn * 64, 0, 0, 0);
}

// The same in kernel X's structure: 256 x 256 x 128 tile, 8 waves (2 x 4, 128 x 64 each), one workgroup per CU, LDS
// A0 | A1 | B0 | B1.  Two blocks of 16 MFMAs (K = 128 each) per K-step; block 0 (m-half 0) carries the reads of
// m-half 1, the barrier sits between the blocks, and block 1 runs column by column so that each B fragment can be
// re-loaded for the next tile as soon as its four MFMAs are out (one B buffer: 128 + 3 x 32 operand registers).
__global__ __launch_bounds__(512) void ca_gemm_fp8_kernel_x(const CaGemmDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  if (!tile_of_block<4, 8>(blockIdx.x, (d.M + XBM - 1) / XBM, (d.N + XBN - 1) / XBN, tm, tn)) return;
  const int m0 = tm * XBM, n0 = tn * XBN;
  KMajorStream<4> la, lb;
  la.init_bytes((const char*)d.A, d.lda, m0, d.M, wave, lane);
  lb.init_bytes((const char*)d.B, d.ldb, n0, d.N, wave, lane);
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int K = d.K;
  constexpr int FBK = 128;
  const int nk = (K + FBK - 1) / FBK;
  auto burst = [&](int kt) {
    if (kt >= nk) return;
    char* na = smem + (kt & 1) * XTILE;
    char* nb = na + 2 * XTILE;
    const bool full = (kt + 1) * FBK <= K;
    if (full) {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        la.issue_one(na, wave, part);
        lb.issue_one(nb, wave, part);
      }
    } else {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        la.issue_one_tail(na, wave, K - kt * FBK, part);
        lb.issue_one_tail(nb, wave, K - kt * FBK, part);
      }
    }
    la.advance();
    lb.advance();
  };
  auto zero_tail = [&](int kt) {
    if (kt != nk - 1 || nk * FBK == K) return;
    char* na = smem + (kt & 1) * XTILE;
    la.zero_fix(na, wave, lane, K - kt * FBK);
    lb.zero_fix(na + 2 * XTILE, wave, lane, K - kt * FBK);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const int g = lane >> 4;
  uint32_t abase[2], bbase[2];  // the two 16-B chunks of this lane's 32-byte operand (fragment 0, stage 0)
  {
    const int ra = wm * 128 + (lane & 15), rb = wn * 64 + (lane & 15);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      abase[h] = lds0 + ra * 128 + (((2 * g + h) ^ ((ra >> 1) & 7)) * 16);
      bbase[h] = lds0 + 2 * XTILE + rb * 128 + (((2 * g + h) ^ ((rb >> 1) & 7)) * 16);
    }
  }
  // operand fragments: [fragment][low / high 16 bytes]
  bf16x8_t A0[4][2], A1[4][2], Bq[4][2];
#define F8X_RD_A(ST, F, dst)                                        \
  do {                                                              \
    dst[0] = lds_read_b128<(ST) * XTILE + (F) * 2048>(abase[0]);    \
    dst[1] = lds_read_b128<(ST) * XTILE + (F) * 2048>(abase[1]);    \
  } while (0)
#define F8X_RD_B(ST, F, dst)                                        \
  do {                                                              \
    dst[0] = lds_read_b128<(ST) * XTILE + (F) * 2048>(bbase[0]);    \
    dst[1] = lds_read_b128<(ST) * XTILE + (F) * 2048>(bbase[1]);    \
  } while (0)
#define F8X_SB __builtin_amdgcn_sched_barrier(0)
  auto pack = [&](bf16x8_t (&f)[2]) {
    const i32x4_t lo = __builtin_bit_cast(i32x4_t, f[0]), hi = __builtin_bit_cast(i32x4_t, f[1]);
    return (fp8x32_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
#define F8X_MM(IH, AF, I, J)                                                                                         \
  do {                                                                                                               \
    acc[(IH) * 4 + (I)][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pack(Bq[J]), pack(AF[I]),              \
                                                                             acc[(IH) * 4 + (I)][J], 0, 0, 0,       \
                                                                             0x7f7f7f7f, 0, 0x7f7f7f7f);             \
    F8X_SB;                                                                                                          \
  } while (0)
  auto wait2 = [&](bf16x8_t (&f)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]),
                   "+v"(f[3][1]));
  };
  auto kstep = [&](auto st_c, int kt) {
    constexpr int ST = decltype(st_c)::value;
    if (wave >= 4) burst(kt + 1);
    // block 0: A (m-half 0) x B; reads A (m-half 1)
    wait2(A0);
    wait2(Bq);
    F8X_SB;
    __builtin_amdgcn_s_setprio(1);
    F8X_MM(0, A0, 0, 0); F8X_MM(0, A0, 0, 1); F8X_RD_A(ST, 4, A1[0]); F8X_SB;
    F8X_MM(0, A0, 0, 2); F8X_MM(0, A0, 0, 3); F8X_RD_A(ST, 5, A1[1]); F8X_SB;
    F8X_MM(0, A0, 1, 0); F8X_MM(0, A0, 1, 1); F8X_RD_A(ST, 6, A1[2]); F8X_SB;
    F8X_MM(0, A0, 1, 2); F8X_MM(0, A0, 1, 3); F8X_RD_A(ST, 7, A1[3]); F8X_SB;
    F8X_MM(0, A0, 2, 0); F8X_MM(0, A0, 2, 1); F8X_MM(0, A0, 2, 2); F8X_MM(0, A0, 2, 3);
    F8X_MM(0, A0, 3, 0); F8X_MM(0, A0, 3, 1); F8X_MM(0, A0, 3, 2); F8X_MM(0, A0, 3, 3);
    __builtin_amdgcn_s_setprio(0);
    wait2(A1);  // the last fragment reads of tile kt have returned
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    zero_tail(kt + 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave < 4) burst(kt + 2);
    F8X_SB;
    // block 1: A (m-half 1) x B, one B column at a time; each column's fragment is re-read for tile kt+1 behind its
    // four MFMAs, the m-half-0 fragments of tile kt+1 ride along (harmless stale data after the last tile)
    __builtin_amdgcn_s_setprio(1);
    F8X_MM(1, A1, 0, 0); F8X_MM(1, A1, 1, 0); F8X_RD_A(1 - ST, 0, A0[0]); F8X_SB;
    F8X_MM(1, A1, 2, 0); F8X_MM(1, A1, 3, 0); F8X_RD_B(1 - ST, 0, Bq[0]); F8X_SB;
    F8X_MM(1, A1, 0, 1); F8X_MM(1, A1, 1, 1); F8X_RD_A(1 - ST, 1, A0[1]); F8X_SB;
    F8X_MM(1, A1, 2, 1); F8X_MM(1, A1, 3, 1); F8X_RD_B(1 - ST, 1, Bq[1]); F8X_SB;
    F8X_MM(1, A1, 0, 2); F8X_MM(1, A1, 1, 2); F8X_RD_A(1 - ST, 2, A0[2]); F8X_SB;
    F8X_MM(1, A1, 2, 2); F8X_MM(1, A1, 3, 2); F8X_RD_B(1 - ST, 2, Bq[2]); F8X_SB;
    F8X_MM(1, A1, 0, 3); F8X_MM(1, A1, 1, 3); F8X_RD_A(1 - ST, 3, A0[3]); F8X_SB;
    F8X_MM(1, A1, 2, 3); F8X_MM(1, A1, 3, 3); F8X_RD_B(1 - ST, 3, Bq[3]); F8X_SB;
    __builtin_amdgcn_s_setprio(0);
  };
  burst(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  zero_tail(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  F8X_RD_B(0, 0, Bq[0]); F8X_RD_B(0, 1, Bq[1]); F8X_RD_B(0, 2, Bq[2]); F8X_RD_B(0, 3, Bq[3]);
  F8X_RD_A(0, 0, A0[0]); F8X_RD_A(0, 1, A0[1]); F8X_RD_A(0, 2, A0[2]); F8X_RD_A(0, 3, A0[3]);
  if (wave < 4) burst(1);
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
  }
#undef F8X_RD_A
#undef F8X_RD_B
#undef F8X_MM
#undef F8X_SB
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  CaGemmDesc dd = d;
  dd.alpha = d.alpha * (d.a_scale ? *d.a_scale : 1.f) * (d.b_scale ? *d.b_scale : 1.f);
#pragma unroll
  for (int ih = 0; ih < 2; ++ih) {
    f32x4_t half[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float rs = 1.f;
      if (d.a_row_scale) {
        const int m = m0 + wm * 128 + (ih * 4 + i) * 16 + (lane & 15);
        rs = d.a_row_scale[m < d.M ? m : d.M - 1];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) half[i][j] = acc[ih * 4 + i][j] * rs;
    }
    gemm_epilogue(dd, half, smem, wave, lane, m0 + wm * 128 + ih * 64, n0 + wn * 64, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done before it is overwritten
  }
}

extern "C" int ca_gemm_fp8(const CaGemmDesc* desc, void* stream) {
  CA_CHECK_ARG(desc != nullptr, "ca_gemm_fp8: null descriptor");
  const CaGemmDesc& d = *desc;
  CA_CHECK_ARG(d.A && d.B && (d.C || d.C2) && d.M > 0 && d.N > 0 && d.K > 0, "ca_gemm_fp8: bad argument");
  CA_CHECK_ARG(d.C8 == nullptr || ((d.epilogue == CA_EPI_GELU || d.epilogue == CA_EPI_DGELU) && d.c8_scale != nullptr && (d.ldc % 8) == 0 &&
                                   ((uintptr_t)d.C8 % 8) == 0),
               "ca_gemm_fp8: C8 needs CA_EPI_GELU / CA_EPI_DGELU, a scale and 8-byte aligned rows");
  CA_CHECK_ARG(d.a_layout == CA_KMAJOR && d.b_layout == CA_KMAJOR && d.batch1 == 1 && d.batch2 == 1 && d.a_kseg == 0 &&
                   d.b_kseg == 0 && !d.a_colsum && !d.c_row_index,
               "ca_gemm_fp8: K-major operands, un-batched, plain rows only");
  CA_CHECK_ARG((d.K % 16) == 0 && (d.lda % 16) == 0 && (d.ldb % 16) == 0 && ((uintptr_t)d.A % 16) == 0 &&
                   ((uintptr_t)d.B % 16) == 0,
               "ca_gemm_fp8: K, lda, ldb must be multiples of 16 bytes and the operands 16-byte aligned");
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)ca_gemm_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr = true;
  }
  // the 256 x 256 kernel where it fills the chip (the bf16 rule), else the 128 x 128 one; ca_gemm_force_kernel(1 / 3)
  // pins either for tests
  const int xtm = (d.M + XBM - 1) / XBM, xtn = (d.N + XBN - 1) / XBN;
  const int64_t xt = (int64_t)xtm * xtn;
  const double xeff = ((double)xt / 256.0) / (double)((xt + 255) / 256);
  const double xfill = ((double)d.M * d.N) / ((double)xtm * XBM * (double)xtn * XBN);
  const bool use_x = g_force_kernel == 3 || (g_force_kernel == 0 && d.K >= 512 && xt >= 160 && xeff * xfill >= 0.70);
  if (use_x) {
    static bool attr_x = false;
    if (!attr_x) {
      hipFuncSetAttribute((const void*)ca_gemm_fp8_kernel_x, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES);
      attr_x = true;
    }
    hipLaunchKernelGGL(ca_gemm_fp8_kernel_x, dim3(tile_grid<4, 8>(xtm, xtn)), dim3(512), X_LDS_BYTES, (hipStream_t)stream, d);
    CA_CHECK_LAUNCH("ca_gemm_fp8");
    return CA_OK;
  }
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  hipLaunchKernelGGL(ca_gemm_fp8_kernel, dim3(tile_grid<8, 8>(ntm, ntn)), dim3(256), LDS_BYTES, (hipStream_t)stream, d);
  CA_CHECK_LAUNCH("ca_gemm_fp8");
  return CA_OK;
}

// Up to four independent GEMMs of the same operand form in one launch of kernel X (grouped launch): used where
// the problems do not fill the chip one by one (the four weight gradients of a transformer layer: 240 + 240 +
// 184 + 64 tiles on 256 CUs at XLS-R-2B, 64 + 64 + 48 + 16 at d = 1024).  All must be un-batched, un-segmented
// and plain (no epilogue, no bias).
// Launch geometry of kernel X.  Default: persistent workgroups (one per CU) with dynamic tile pulls whenever the tile
// grid is larger than the chip and un-batched; CA_X_PERSIST=0 restores one workgroup per tile.
#define X_LAUNCH_LDS (CA_X_SLAB ? X_SLAB_LDS : X_LDS_BYTES + 64)
static void x_launch_geometry(CaGemmGroup& g, unsigned vgrid, unsigned nbz, dim3& grid) {
  static const int persist = [] { const char* e = getenv("CA_X_PERSIST"); return e ? atoi(e) : 1; }();
  const unsigned ncu = x_compute_cus();
  static unsigned seq = 0;
  g.vgrid = (int)vgrid;
  g.cnt = nullptr;
  g.dyn_first = g_compute_cus > 0 ? 1 : 0;
  grid = dim3(vgrid, 1, nbz);
  if (persist && nbz == 1 && vgrid > ncu) {
    unsigned* base = nullptr;
    if (hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_x_cnt)) == hipSuccess && base) {
      g.cnt = base + (size_t)(seq++ % X_CNT_SLOTS) * 8;
      grid.x = ncu;
    }
  }
}

extern "C" int ca_gemm_bf16_group(const CaGemmDesc* descs, int32_t count, void* stream) {
  CA_CHECK_ARG(descs && count >= 1 && count <= X_GROUP_MAX, "ca_gemm_bf16_group: 1..8 problems");
  CaGemmGroup g;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    const CaGemmDesc* p = descs + i;
    CA_CHECK_ARG(p->A && p->B && p->C && p->M > 0 && p->N > 0 && p->K >= 64, "ca_gemm_bf16_group: bad problem %d", i);
    CA_CHECK_ARG(p->batch1 == 1 && p->batch2 == 1 && p->a_kseg == 0 && p->b_kseg == 0 && p->epilogue == CA_EPI_NONE &&
                     p->bias == nullptr && p->dropout_p == 0.f,
                 "ca_gemm_bf16_group: only plain un-batched problems");
    CA_CHECK_ARG((p->lda % 8) == 0 && (p->ldb % 8) == 0 && ((uintptr_t)p->A % 16) == 0 && ((uintptr_t)p->B % 16) == 0,
                 "ca_gemm_bf16_group: alignment");
    CA_CHECK_ARG(p->a_layout == descs->a_layout && p->b_layout == descs->b_layout,
                 "ca_gemm_bf16_group: the problems differ in operand form");
    CA_CHECK_ARG(p->c_sumsq == nullptr || p->epilogue == CA_EPI_NONE, "ca_gemm_bf16_group: c_sumsq needs a plain epilogue");
    g.d[i] = *p;
    g.first[i] = total;
    total += ((p->M + XBM - 1) / XBM) * ((p->N + XBN - 1) / XBN);
  }
  g.total = total;
  for (int i = count; i < X_GROUP_MAX; ++i) {
    g.d[i] = descs[0];
    g.first[i] = total;
  }
  g.count = count > 1 ? count : 0;
  static bool attr = false;
#define XK(x, y) ca_gemm_kernel_x<x, y, false>
  if (!attr) {
    const void* fs[4] = {(const void*)XK(0, 0), (const void*)XK(0, 1), (const void*)XK(1, 0), (const void*)XK(1, 1)};
    for (int i = 0; i < 4; ++i) hipFuncSetAttribute(fs[i], hipFuncAttributeMaxDynamicSharedMemorySize, X_LAUNCH_LDS);
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid, block(512);
  x_launch_geometry(g, (unsigned)(count > 1 ? 8 * ((total + 7) / 8)
                                            : tile_grid<4, 8>((descs->M + XBM - 1) / XBM, (descs->N + XBN - 1) / XBN)), 1, grid);
  const int lay = (descs->a_layout ? 2 : 0) + (descs->b_layout ? 1 : 0);
  ProfRec r;
  if (g_prof_on) {  // live roofline timing (bench.py): the group counts as one launch of kernel X
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    r.flops = 0.0;
    for (int i = 0; i < count; ++i) r.flops += 2.0 * descs[i].M * (double)descs[i].N * descs[i].K;
    r.variant = 2 * 8 + lay;
    g_prof_e0 = r.e0;
    g_prof_e1 = r.e1;
  }
  switch (lay) {
    case 0: CA_LAUNCH((XK(0, 0)), grid, block, X_LAUNCH_LDS, s, g); break;
    case 1: CA_LAUNCH((XK(0, 1)), grid, block, X_LAUNCH_LDS, s, g); break;
    case 2: CA_LAUNCH((XK(1, 0)), grid, block, X_LAUNCH_LDS, s, g); break;
    default: CA_LAUNCH((XK(1, 1)), grid, block, X_LAUNCH_LDS, s, g); break;
  }
#undef XK
  if (g_prof_on) {
    g_prof_e0 = g_prof_e1 = nullptr;
    g_prof.push_back(r);
  }
  CA_CHECK_LAUNCH("ca_gemm_bf16_group");
  return CA_OK;
}

static int ca_gemm_launch(const CaGemmDesc* desc, void* stream) {
  CA_CHECK_ARG(desc != nullptr, "ca_gemm_bf16: null descriptor");
  // (a copy: the library, not the caller, owns xcd_balanced - every XCD gets the same number of tiles whenever the
  // host said that the chip is shared with a resident kernel, ca_gemm_set_compute_cus)
  CaGemmDesc dcopy = *desc;
  dcopy.xcd_balanced = g_compute_cus > 0 ? 1 : 0;
  const CaGemmDesc& d = dcopy;
  const bool bal = dcopy.xcd_balanced != 0;
  CA_CHECK_ARG(d.A && d.B &&
                   (d.C || ((d.epilogue == CA_EPI_GELU || d.epilogue == CA_EPI_GELU_RESIDUAL) && d.C2)),
               "ca_gemm_bf16: null operand");
  CA_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "ca_gemm_bf16: bad shape %d %d %d", d.M, d.N,
               d.K);
  CA_CHECK_ARG(d.batch1 > 0 && d.batch2 > 0, "ca_gemm_bf16: bad batch");
  CA_CHECK_ARG((d.lda % 8) == 0 && (d.ldb % 8) == 0, "ca_gemm_bf16: lda/ldb must be multiples of 8");
  CA_CHECK_ARG((d.sA1 % 8) == 0 && (d.sA2 % 8) == 0 && (d.sB1 % 8) == 0 && (d.sB2 % 8) == 0,
               "ca_gemm_bf16: batch strides must be multiples of 8");
  CA_CHECK_ARG(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0,
               "ca_gemm_bf16: A/B must be 16-byte aligned");
  CA_CHECK_ARG(d.a_kseg >= 0 && d.b_kseg >= 0, "ca_gemm_bf16: negative kseg");
  if (d.epilogue == CA_EPI_RESIDUAL || d.epilogue == CA_EPI_DGELU || d.epilogue == CA_EPI_GELU_RESIDUAL)
    CA_CHECK_ARG(d.R != nullptr, "ca_gemm_bf16: epilogue needs R");
  CA_CHECK_ARG(d.dropout_p >= 0.f && d.dropout_p < 1.f, "ca_gemm_bf16: bad dropout_p");
  CA_CHECK_ARG(d.C8 == nullptr || ((d.epilogue == CA_EPI_GELU || d.epilogue == CA_EPI_DGELU) && d.c8_scale != nullptr && d.batch1 == 1 && d.batch2 == 1 &&
                                   (d.ldc % 8) == 0 && ((uintptr_t)d.C8 % 8) == 0 && d.M > 32),
               "ca_gemm_bf16: C8 needs CA_EPI_GELU / CA_EPI_DGELU, a scale, an un-batched problem and 8-byte aligned rows");
  if (d.a_colsum)
    CA_CHECK_ARG(d.a_layout == CA_MNMAJOR && d.b_layout == CA_MNMAJOR && d.batch1 == 1 && d.batch2 == 1 && d.a_kseg == 0,
                 "ca_gemm_bf16: a_colsum needs the un-batched weight-gradient form");
  if (d.c_sumsq)
    CA_CHECK_ARG(d.epilogue == CA_EPI_NONE && d.batch1 == 1 && d.batch2 == 1 && d.c_row_index == nullptr && d.c_split_n == 0,
                 "ca_gemm_bf16: c_sumsq needs an un-batched output with the plain epilogue");

  hipStream_t s = (hipStream_t)stream;
  const int lay = (d.a_layout ? 2 : 0) + (d.b_layout ? 1 : 0);
  CA_CHECK_ARG(d.a_layout == CA_KMAJOR || d.a_layout == CA_MNMAJOR, "ca_gemm_bf16: bad a_layout");
  CA_CHECK_ARG(d.b_layout == CA_KMAJOR || d.b_layout == CA_MNMAJOR, "ca_gemm_bf16: bad b_layout");
  // Skinny M (greedy decoding: one token per clip): weight streaming without LDS staging.
  // (33 .. 128 rows take it when the caller asks for what only this form has - the K|V-cache row scatter of a decoded
  // token - or when the problem is too narrow to give the tiled kernels a grid: N < 8192 means <= 64 tiles of 128 x 128)
  const bool skinny_wide = d.M > 32 && d.M <= 128 && d.C8 == nullptr && d.c_sumsq == nullptr &&
                           (d.a_ln_gamma || d.c_row_index || d.c_split_n > 0 || d.N <= 8192 || d.M <= 64);
  if (g_force_kernel == 0 && (d.M <= 32 || skinny_wide) && d.a_layout == CA_KMAJOR && d.b_layout == CA_KMAJOR && d.batch1 == 1 &&
      d.batch2 == 1 && d.a_kseg == 0 && d.b_kseg == 0 && d.dropout_p == 0.f && d.epilogue != CA_EPI_DGELU) {
    g_last_kind = 0;
    CA_CHECK_ARG(d.c_split_n == 0 || (d.C_hi && (d.c_split_n % 16) == 0 && d.c_split_n < d.N && d.epilogue == CA_EPI_NONE),
                 "ca_gemm_bf16: c_split_n needs C_hi, a multiple of 16 below N and no epilogue");
    unsigned gy = d.M <= 32 ? 1u : (unsigned)((d.M + 31) / 32);  // row blocks of 32 (blockIdx.y)
    // 17 .. 128 rows: 16-row workgroups over blockIdx.y wherever that leaves the launch at most four workgroups per CU
    // (every projection of a decoder layer; not the vocabulary).  The kernel's pace is set by the requests its waves keep
    // in flight, not by bytes: at 32 clips the 32-row form (one workgroup per column block, two row blocks against the
    // same weight fragment - round 3) gave N = 1024 launches 64 workgroups; 16-row workgroups re-read the weights from
    // L2 but double the waves: 3.06 -> 2.66 ms per token at 32 clips, 4.50 -> 3.90 at 64, 6.96 -> 6.60 at 128 (round 5;
    // 64-row workgroups, the opposite direction, measured 4.40 at 64).  Same K split per output element: same bits.
    // CA_SKINNY_MB1=0 restores 32-row workgroups.
    static const int mb1 = [] { const char* e = getenv("CA_SKINNY_MB1"); return e ? atoi(e) : 4; }();
    static const int mb1_rows = [] { const char* e = getenv("CA_SKINNY_MB1_ROWS"); return e ? atoi(e) : 16; }();
    const bool rows16 = mb1 && d.M > mb1_rows && d.M > 16 && (unsigned)((d.N + 15) / 16) * gy <= (unsigned)mb1 * x_device_cus();
    if (rows16) gy = (unsigned)((d.M + 15) / 16);
    // columns per workgroup (round 6): the widest of 16 / 8 / 4 that still gives the launch 3/4 of a workgroup per CU
    // (16-row workgroups only; CA_SKINNY_NT=16 restores sixteen everywhere)
    static const int nt_force = [] { const char* e = getenv("CA_SKINNY_NT"); return e ? atoi(e) : 0; }();
    int nt = 16;
    if (d.M <= 16 || rows16) {
      const unsigned want = 3u * x_device_cus() / 4u;
      if ((unsigned)((d.N + 15) / 16) * gy < want) nt = (unsigned)((d.N + 7) / 8) * gy >= want ? 8 : 4;
      if (nt_force == 4 || nt_force == 8 || nt_force == 16) nt = nt_force;
    }
    const dim3 grid((unsigned)((d.N + nt - 1) / nt), gy);
    if (d.a_ln_gamma) {
      CA_CHECK_ARG(d.a_ln_beta && d.K <= 2048 && ((uintptr_t)d.a_ln_gamma % 16) == 0 && ((uintptr_t)d.a_ln_beta % 16) == 0,
                   "ca_gemm_bf16: a_ln_gamma needs a_ln_beta, K <= 2048 and 16-byte aligned vectors");
      const int mb = (d.M <= 16 || rows16) ? 1 : 2;  // (33 .. 128 rows: 32- or 16-row blocks over blockIdx.y)
      const int nch = (d.K + 511) / 512 <= 2 ? 2 : (d.K + 511) / 512;
      const size_t lds = (size_t)4 * mb * 256 * sizeof(float) + (size_t)16 * mb * (d.K + SKINNY_LN_PAD) * 2;
#define SKINNY_LN(MBV, NCHV, NTV)                                                                                    \
  do {                                                                                                               \
    static bool attr_ln = false;                                                                                     \
    if (!attr_ln) {                                                                                                  \
      hipFuncSetAttribute((const void*)ca_gemm_skinny_kernel<MBV, NCHV, 8, NTV>,                                     \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                   \
      attr_ln = true;                                                                                                \
    }                                                                                                                \
    CA_LAUNCH((ca_gemm_skinny_kernel<MBV, NCHV, 8, NTV>), grid, dim3(256), lds, s, d);                               \
  } while (0)
#define SKINNY_LN1(NCHV)                   \
  do {                                     \
    if (nt == 16) SKINNY_LN(1, NCHV, 16);  \
    else if (nt == 8) SKINNY_LN(1, NCHV, 8); \
    else SKINNY_LN(1, NCHV, 4);            \
  } while (0)
      if (mb == 1 && nch == 2) SKINNY_LN1(2);
      else if (mb == 1 && nch == 3) SKINNY_LN1(3);
      else if (mb == 1) SKINNY_LN1(4);
      else if (nch == 2) SKINNY_LN(2, 2, 16);
      else if (nch == 3) SKINNY_LN(2, 3, 16);
      else SKINNY_LN(2, 4, 16);
#undef SKINNY_LN1
#undef SKINNY_LN
    } else if (d.M <= 16 || rows16) {
      // k-steps in flight per wave: its whole K quarter up to 32 (K = 4096: one round of loads instead of four;
      // CA_SKINNY_U=8 restores eight)
      static const int u_force = [] { const char* e = getenv("CA_SKINNY_U"); return e ? atoi(e) : 0; }();
      const int per = ((d.K + 31) / 32 + 3) / 4;
      int u = per <= 8 ? 8 : (per <= 16 ? 16 : 32);
      if (u_force == 8 || u_force == 16 || u_force == 32) u = u_force;
#define SKINNY_P(UV, NTV) CA_LAUNCH((ca_gemm_skinny_kernel<1, 0, UV, NTV>), grid, dim3(256), 4 * 256 * sizeof(float), s, d)
#define SKINNY_PU(UV)            \
  do {                           \
    if (nt == 16) SKINNY_P(UV, 16); \
    else if (nt == 8) SKINNY_P(UV, 8); \
    else SKINNY_P(UV, 4);        \
  } while (0)
      if (u == 8) SKINNY_PU(8);
      else if (u == 16) SKINNY_PU(16);
      else SKINNY_PU(32);
#undef SKINNY_PU
#undef SKINNY_P
    } else {
      CA_LAUNCH((ca_gemm_skinny_kernel<2, 0>), grid, dim3(256), 8 * 256 * sizeof(float), s, d);
    }
    CA_CHECK_LAUNCH("ca_gemm_bf16");
    return CA_OK;
  }
  CA_CHECK_ARG(!d.a_ln_gamma, "ca_gemm_bf16: a_ln_gamma exists in the skinny form only (M <= 128, K-major operands, un-batched)");
  CA_CHECK_ARG(!d.c_row_index && d.c_split_n == 0,
               "ca_gemm_bf16: c_row_index / c_split_n exist in the skinny form only (M <= 128, K-major operands, un-batched)");
  // Kernel choice: the 256x128 pipelined kernel runs one workgroup per CU, so it needs enough
  // tiles to fill the chip; small or heavily batched problems use the 128x128 kernel.
  const int64_t nb = (int64_t)d.batch1 * d.batch2;
  const int64_t tiles_l = (int64_t)((d.M + LBM - 1) / LBM) * ((d.N + LBN - 1) / LBN) * nb;
  // Measured on MI355X (profiles/r01_gemm_shapes.txt): at the path's shapes (K = 1920..7680, M = 3992)
  // the 128x128 kernel with two workgroups per CU equals or beats the 256x128 one-per-CU kernel,
  // because its second workgroup hides the epilogue; the L kernel is kept selectable for tuning.
  int use_l = (g_force_kernel == 2 && d.a_kseg == 0 && d.b_kseg == 0) ? 1 : 0;
  // Kernel L (256x128, three-stage ring) takes the shapes kernel X does not fill and that give it 160 .. 768 tiles
  // (0.4 .. 3 rounds of one workgroup per CU; CA_GEMM_L_MIN, default 100 tiles: from there it also beats the 128x128
  // kernel on the d = 1024 models, XLS-R-300M step 19.95 -> 19.1 ms): the N = d projections and data gradients and q|k|v at the 2B shape.  Its
  // two-tiles-ahead LDS-DMA keeps it fed under the optimiser's HBM traffic, where the 128x128 kernel (one tile ahead)
  // loses 25 %: XLS-R-2B step 79.2 -> 76.5 ms on one box (tools/archive/exp_l3.sh).  CA_GEMM_PREFER_L=0 turns it off, a larger
  // value widens the tile-count window (x 256).
  static const int prefer_l = [] { const char* e = getenv("CA_GEMM_PREFER_L"); return e ? atoi(e) : 3; }();
  // Kernel X (256x256): only where it fills the chip -- at least ~0.7 tiles per CU in its last round.
  const int xtm = (d.M + XBM - 1) / XBM, xtn = (d.N + XBN - 1) / XBN;
  const int64_t xt = (int64_t)xtm * xtn * nb;
  const int64_t cus = (int64_t)x_compute_cus();  // (256 on MI355X; fewer beside a resident collective: ca_gemm_set_compute_cus)
  const double xwaves = (double)xt / (double)cus;
  const double xeff = xwaves / (double)((xt + cus - 1) / cus);                   // last-wave occupancy
  const double xfill = ((double)d.M * d.N) / ((double)xtm * XBM * (double)xtn * XBN);  // tile padding waste
  // The MN-major x MN-major (weight-gradient) form gains most from the 256x256 tile (1.0 PFLOP/s against 0.63
  // for S inside the training step), so it switches at a lower fill than the other forms.
  const bool tn = d.a_layout == CA_MNMAJOR && d.b_layout == CA_MNMAJOR;
  // (thresholds from tools/dev_gemm_rule.py on the models' shapes: X wins from ~73 % occupancy of its last round
  // - 188 / 192 / 564 tiles - and loses at 68 % - 368 tiles)
  int use_x = (g_force_kernel == 0 && d.K >= 512 && xt >= 160 && xeff * xfill >= (tn ? 0.60 : 0.70)) ? 1 : 0;
  if (g_force_kernel == 3 || d.a_colsum) use_x = 1;  // the column sums live in kernel X only
  static const int l_over_x = [] { const char* e = getenv("CA_GEMM_L_OVER_X"); return e ? atoi(e) : 0; }();
  static const int l_min = [] { const char* e = getenv("CA_GEMM_L_MIN"); return e ? atoi(e) : 100; }();
  if (prefer_l && g_force_kernel == 0 && d.a_kseg == 0 && d.b_kseg == 0 && d.K >= 512 && tiles_l >= l_min &&
      (tiles_l <= cus * prefer_l || (use_x && l_over_x)) && !d.a_colsum &&
      (!use_x || (l_over_x == 1 && !tn) || l_over_x == 2 || (l_over_x == 3 && lay == 0))) {
    use_l = 1;
    use_x = 0;
  }
  // Beside a resident kernel that holds some CUs (ca_gemm_set_compute_cus(n), n below the chip's count) the rules above -
  // tuned for exactly 256 CUs - pick single-round tilings that then run TWO rounds (240 tiles on 224 CUs).  There the
  // choice is made by counting rounds on the CUs that are left: cost = rounds x tile work / the shape's efficiency, in
  // units of one 128 x 128 tile's work (kernel S: two tiles per CU at a time; efficiencies from
  // profiles/r05_gemm_shapes.txt, weight-gradient form in brackets): X 4 / 1.0, L 2 / 0.93 [0.85], S 2 / 0.8 [0.6] per
  // pair, M 1 / 0.6.  tenant_kind: -1 = not in this regime.
  int tenant_kind = -1;
  if (g_force_kernel == 0 && g_compute_cus > 0 && cus < (int64_t)x_device_cus() && nb == 1 && d.a_kseg == 0 && d.b_kseg == 0 &&
      d.K >= 512 && !d.a_colsum && d.M > 128) {
    const int64_t ts = (int64_t)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    auto rounds = [](int64_t tiles, int64_t slots) { return (double)((tiles + slots - 1) / slots); };
    const double cx = rounds(xt, cus) * 4.0, cl = rounds(tiles_l, cus) * 2.0 / (tn ? 0.85 : 0.93),
                 cs = rounds(ts, 2 * cus) * 2.0 / (tn ? 0.6 : 0.8), cm = rounds(ts, cus) * 1.0 / 0.6;
    tenant_kind = 2;
    double best = cx;
    if (cl < best) { best = cl; tenant_kind = 1; }
    if (cs < best) { best = cs; tenant_kind = 0; }
    if (cm < best) { best = cm; tenant_kind = 3; }
    use_x = tenant_kind == 2;
    use_l = tenant_kind == 1;
  }
  g_last_kind = use_x ? 2 : (use_l ? 1 : 0);
  if (use_x) {
    static bool xattr = false;
    const bool ks = d.a_kseg > 0 || d.b_kseg > 0;
#define XK(a, b, k) ca_gemm_kernel_x<a, b, k>
    if (!xattr) {
      const void* fs[8] = {(const void*)XK(0, 0, false), (const void*)XK(0, 1, false), (const void*)XK(1, 0, false),
                           (const void*)XK(1, 1, false), (const void*)XK(0, 0, true),  (const void*)XK(0, 1, true),
                           (const void*)XK(1, 0, true),  (const void*)XK(1, 1, true)};
      for (int i = 0; i < 8; ++i) hipFuncSetAttribute(fs[i], hipFuncAttributeMaxDynamicSharedMemorySize, X_LAUNCH_LDS);
      xattr = true;
    }
    dim3 grid, block(512);
    CaGemmGroup one;
    one.d[0] = d;
    one.count = 0;
    one.first[0] = 0;
    one.total = 0;
    x_launch_geometry(one, tile_grid<4, 8>(xtm, xtn, bal), (unsigned)nb, grid);
    switch (lay + (ks ? 4 : 0)) {
      case 0: CA_LAUNCH((XK(0, 0, false)), grid, block, X_LAUNCH_LDS, s, one); break;
      case 1: CA_LAUNCH((XK(0, 1, false)), grid, block, X_LAUNCH_LDS, s, one); break;
      case 2: CA_LAUNCH((XK(1, 0, false)), grid, block, X_LAUNCH_LDS, s, one); break;
      case 3: CA_LAUNCH((XK(1, 1, false)), grid, block, X_LAUNCH_LDS, s, one); break;
      case 4: CA_LAUNCH((XK(0, 0, true)), grid, block, X_LAUNCH_LDS, s, one); break;
      case 5: CA_LAUNCH((XK(0, 1, true)), grid, block, X_LAUNCH_LDS, s, one); break;
      case 6: CA_LAUNCH((XK(1, 0, true)), grid, block, X_LAUNCH_LDS, s, one); break;
      default: CA_LAUNCH((XK(1, 1, true)), grid, block, X_LAUNCH_LDS, s, one); break;
    }
#undef XK
    CA_CHECK_LAUNCH("ca_gemm_bf16");
    return CA_OK;
  }
  if (use_l) {
    static bool attr_done = false;
    if (!attr_done) {
      hipFuncSetAttribute((const void*)ca_gemm_kernel_l<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS_BYTES);
      hipFuncSetAttribute((const void*)ca_gemm_kernel_l<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS_BYTES);
      hipFuncSetAttribute((const void*)ca_gemm_kernel_l<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS_BYTES);
      hipFuncSetAttribute((const void*)ca_gemm_kernel_l<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS_BYTES);
      attr_done = true;
    }
    dim3 grid(tile_grid<4, 8>((d.M + LBM - 1) / LBM, (d.N + LBN - 1) / LBN, bal), 1, (unsigned)nb);
    dim3 block(512);
    switch (lay) {
      case 0: CA_LAUNCH((ca_gemm_kernel_l<0, 0>), grid, block, L_LDS_BYTES, s, d); break;
      case 1: CA_LAUNCH((ca_gemm_kernel_l<0, 1>), grid, block, L_LDS_BYTES, s, d); break;
      case 2: CA_LAUNCH((ca_gemm_kernel_l<1, 0>), grid, block, L_LDS_BYTES, s, d); break;
      default: CA_LAUNCH((ca_gemm_kernel_l<1, 1>), grid, block, L_LDS_BYTES, s, d); break;
    }
  } else {
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    dim3 grid(tile_grid<8, 8>(ntm, ntn, bal), 1, (unsigned)nb);
    dim3 block(256);
    const size_t lds = LDS_BYTES;
    const bool ks = d.a_kseg > 0 || d.b_kseg > 0;
    // at most one tile per CU: kernel M (two waves per SIMD on the same tile; CA_GEMM_M=0 switches it off)
    static const int m_max = [] { const char* e = getenv("CA_GEMM_M"); return e ? atoi(e) : 256; }();
    if (!ks && (g_force_kernel == 5 || tenant_kind == 3 ||
                (g_force_kernel == 0 && tenant_kind < 0 && (int64_t)ntm * ntn * nb <= m_max && d.K >= 2 * BK))) {
      static bool mattr = false;
      if (!mattr) {
        const void* fs[4] = {(const void*)ca_gemm_kernel_m<0, 0>, (const void*)ca_gemm_kernel_m<0, 1>,
                             (const void*)ca_gemm_kernel_m<1, 0>, (const void*)ca_gemm_kernel_m<1, 1>};
        for (int i = 0; i < 4; ++i) hipFuncSetAttribute(fs[i], hipFuncAttributeMaxDynamicSharedMemorySize, M_LDS_BYTES);
        mattr = true;
      }
      const dim3 mblock(512);
      switch (lay) {
        case 0: CA_LAUNCH((ca_gemm_kernel_m<0, 0>), grid, mblock, M_LDS_BYTES, s, d); break;
        case 1: CA_LAUNCH((ca_gemm_kernel_m<0, 1>), grid, mblock, M_LDS_BYTES, s, d); break;
        case 2: CA_LAUNCH((ca_gemm_kernel_m<1, 0>), grid, mblock, M_LDS_BYTES, s, d); break;
        default: CA_LAUNCH((ca_gemm_kernel_m<1, 1>), grid, mblock, M_LDS_BYTES, s, d); break;
      }
      CA_CHECK_LAUNCH("ca_gemm_bf16");
      return CA_OK;
    }
    switch (lay + (ks ? 4 : 0)) {
      case 0: CA_LAUNCH((ca_gemm_kernel<0, 0, false>), grid, block, lds, s, d); break;
      case 1: CA_LAUNCH((ca_gemm_kernel<0, 1, false>), grid, block, lds, s, d); break;
      case 2: CA_LAUNCH((ca_gemm_kernel<1, 0, false>), grid, block, lds, s, d); break;
      case 3: CA_LAUNCH((ca_gemm_kernel<1, 1, false>), grid, block, lds, s, d); break;
      case 4: CA_LAUNCH((ca_gemm_kernel<0, 0, true>), grid, block, lds, s, d); break;
      case 5: CA_LAUNCH((ca_gemm_kernel<0, 1, true>), grid, block, lds, s, d); break;
      case 6: CA_LAUNCH((ca_gemm_kernel<1, 0, true>), grid, block, lds, s, d); break;
      default: CA_LAUNCH((ca_gemm_kernel<1, 1, true>), grid, block, lds, s, d); break;
    }
  }
  CA_CHECK_LAUNCH("ca_gemm_bf16");
  return CA_OK;
}
