// Greedy decoding: one persistent launch per token for a whole Whisper decoder (include/coral_amd.h,
// ca_whisper_decode_token).  gfx950 only.
//
// Why one launch: a decoded token at 8 .. 16 clips is a chain of ~170 dependent launches, and a dependent launch on this
// chip costs ~4.7 us from dispatch to completion before it moves a byte (profiles/r06_token16_base_by_grid.txt: the
// embedding gather, 8 workgroups, takes 4.7 us) - 0.8 ms of the 2.0 ms a token took, in which HBM idles.  An all-to-all
// seam INSIDE a launch costs ~3.2 us (tools/r06/seam_bench.hip), and everything that does not depend on the token's
// activations - the weights - can be in LDS before the seam that needs it.
//
// Structure.  G = one 256-thread workgroup per CU, all resident (~156 KB of LDS each).  A layer is eight phases:
//   A  LayerNorm + q|k|v projection (K|V rows into the self-attention cache)      all-to-all in
//   B  self-attention over the cache, one (clip, head) per workgroup
//   C  out-projection + residual                                                  all-to-all in
//   D  LayerNorm + cross-attention query projection                               all-to-all in
//   E  attention over the cached encoder K|V, (clip, head, key split) per workgroup: the token's HBM stream
//   F  out-projection + residual                                                  all-to-all in
//   G  LayerNorm + fc1 + GELU                                                     all-to-all in
//   H  fc2 + residual                                                             all-to-all in
// then LayerNorm + the tied output projection (16 vocabulary rows per tile, tiles dealt round-robin) and the greedy pick.
// Projection phases give every workgroup N / G output columns (groups of four: 4 at N = 1024) against the WHOLE K: a
// wave owns a quarter of K, as in ca_gemm_skinny_kernel, so per output element the MFMA chain and the (p0 + p1) + (p2 + p3)
// combine are that kernel's - the same bits.  The residual stream's columns of a workgroup stay in registers.
//
// Seams.  Every workgroup owns one progress word (flags[w] = phases it has completed; kept in 8 replicas, each polled by
// an eighth of the workgroups, all stored by one wave instruction of the owner).  A phase's outputs are stored
// write-through (sc1) by wave 0, which then drains (s_waitcnt vmcnt(0)) and stores the word (sc1); a consumer's wave 0
// polls ALL words with one 16-byte sc1 load per lane, then the workgroup's barrier, then every load of handed-off bytes is
// an sc1 load to registers (MI355X_MICROARCH.md, visibility: the form "one lane of each storing workgroup / sc1 poll of
// every shard / barrier / sc1 loads").  Phase B's V tiles arrive by LDS-DMA: the one row of them written in this launch is
// fetched again by an sc1 load and written over the tile's copy in LDS (dk_attend, fresh_row).  Every spin is bounded: on a timeout the workgroup sets status[0] and leaves, the others follow.
// Buffers are written once per layer and read in the next phase, so re-use one layer later is ordered by the seams.
//
// Weights.  A wave's quarter of K of its workgroup's columns of every matrix goes through a private LDS ring (20 pieces of
// 1 KiB) by LDS-DMA, issued as far ahead as the ring holds (about one layer): piece = 4 columns x 128 k, 256 contiguous
// bytes per column, chunk positions XOR-ed so that the MFMA fragment reads (ds_read_b128) hit 16 distinct bank slots.
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* dk_lptr_t;
typedef __attribute__((ext_vector_type(8))) short dk_s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned dk_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned dk_u32x2;

__device__ __attribute__((aligned(16))) uint32_t g_dec_zero_page[4];

#define DK_NEG_BIG (-1.0e30f)
#define DK_LOG2E 1.44269504088896340736f
#define DK_RP 20             // pieces (1 KiB) of a wave's weight ring
#define DK_SCRATCH (64 * 1024)  // LayerNorm image | V rings + merge buffers
#define DK_PART (14 * 1024)   // partial tiles (4 KiB) | seam word, kinds (1 KiB) | the layer records (7 KiB: 44 layers) | biases (2 KiB)
#define DK_MAXLAYERS 44
#define DK_LDS (DK_PART + DK_SCRATCH + 4 * DK_RP * 1024)
#define DK_XPAD 32           // bf16 elements between rows of the LayerNorm image beyond K (ca_gemm_skinny_kernel's pad)
#define DK_SPLIT_ROW 66      // floats of a partial: m, l, 64 output columns
#define DK_MAXTILES 16       // vocabulary tiles per workgroup (V <= 16 * 16 * G)
#define DK_SPIN_LIMIT (1u << 21)
#define DK_FLAG_STRIDE 256   // words between replicas of the progress words (one replica = 1 KiB: 256 workgroups)
#define DK_FLAG_REPS_MAX 8

// CaDecodeLayer as 20 64-bit words (the kernel keeps the records in LDS: a pointer is one ds_read away, not a scalar load
// from device memory in front of every phase)
enum { LY_LN1G, LY_LN1B, LY_WQKV, LY_BQKV, LY_WO, LY_BO, LY_LN2G, LY_LN2B, LY_WQ2, LY_BQ2, LY_WO2, LY_BO2, LY_LN3G, LY_LN3B,
       LY_W1, LY_B1, LY_W2, LY_B2, LY_SELFKV, LY_CROSSKV, LY_WORDS };
static_assert(sizeof(CaDecodeLayer) == LY_WORDS * 8, "CaDecodeLayer is 20 pointers");
struct DecArgs {
  const CaDecodeLayer* layers;
  int n_layers, B, d, f, H, Te, Lmax, V;
  CaKeySplit split;  // of the cross-attention's (clip, head) items
  const unsigned short *embed, *pos_tab;
  const float *lnf_g, *lnf_b;
  float eps, scale;
  float* logits;
  int64_t ld_logits;
  const uint8_t* suppress;
  int32_t* out;
  uint8_t* done;
  int64_t* ids;
  int64_t ld_ids;
  int32_t *tok, *pos, *klen;
  int32_t pad, eos;
  // workspace
  unsigned* flags;          // [flag_reps][DK_FLAG_STRIDE] progress words
  int flag_reps;            // 1, 2, 4 or 8: workgroup w polls replica w & (flag_reps - 1)
  int poll_sleep;
  unsigned* split_cnt;  // [n_layers][B * H]
  unsigned short *q, *ctx, *h1, *q2, *ctx2, *h2, *h, *gbuf;
  float* slab;          // [B * H * ns][DK_SPLIT_ROW]
  float* amax_val;      // [G][16]
  int* amax_idx;
  unsigned* status;
  unsigned long long* stamps;  // debug (ca_debug_decode_stamps): [G][stamp_nph][2] realtime ticks, or NULL
  int stamp_nph, stamp_tid, stamp_fine;
  int pre_issue;  // phase E: the first K|V tiles are asked for before the seam
  int ring_budget;
  int cross_hm;  // the encoder K|V are head-major copies
  int b_fence;
};

// ---- memory helpers ------------------------------------------------------------------------------------------------
// Loads whose result an asm statement defines: the compiler knows nothing of the latency, so a use must be ordered behind
// the wait by a data dependence (dk_tie after dk_vm0; cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dk_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void dk_tie(u16x8_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void dk_tie(bf16x8_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void dk_tie(dk_u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void dk_tie(f32x4_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ u16x8_t dk_ld16_sc1(const void* p) {
  u16x8_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ u16x8_t dk_ld16_nt(const void* p) {
  u16x8_t v;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ bf16x8_t dk_ld16(const void* p) {
  bf16x8_t v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ dk_u32x4 dk_ld16u_sc1(const void* p) {
  dk_u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void dk_st8_sc1(void* p, dk_u32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void dk_st4_sc1(void* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void dk_st2_sc1(void* p, unsigned v) {
  asm volatile("global_store_short %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// LDS-DMA: 16 bytes per lane from `g` to lds_wave_base + 16 * lane
__device__ __forceinline__ void dk_glds16(const void* g, uint32_t lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base))
               : "memory", "m0");
}
// the same with the non-temporal policy: bytes ONE workgroup reads once per token (its weight columns, its K|V strip)
__device__ __forceinline__ void dk_glds16_nt(const void* g, uint32_t lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
               :
               : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base))
               : "memory", "m0");
}
// Which streams carry it (bit 0: weight ring, bit 1: V tiles, bit 2: K fragments of the cross-attention).  Measured per token
// at 16 | 8 clips (same box, tools/r06/token_step_time.py): none 1 486 | 1 242 us, ring only 1 489 | 1 233, V tiles 1 469 | 1 223,
// V tiles + cross K 1 464 | 1 226, all three 1 465 | 1 230: the K|V strips (98 MB per layer at 16 clips) no longer displace
// what the L2s and the Infinity Cache hold for the phases around them; the ring's pieces gain nothing.
#ifndef DK_NT
#define DK_NT 6
#endif
__device__ __forceinline__ uint32_t dk_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(dk_lptr_t)p; }
__device__ __forceinline__ bf16x8_t dk_zero8() { return __builtin_bit_cast(bf16x8_t, (f32x4_t){0.f, 0.f, 0.f, 0.f}); }

// wait until at most n vector-memory operations of this wave are outstanding.  Loads complete in order, so "at most n"
// means every load older than the n youngest operations has landed (stores in flight only make the wait stricter).
__device__ __forceinline__ void dk_wait_vm(int n) {
  switch (n) {
#define DK_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    DK_W(1) DK_W(2) DK_W(3) DK_W(4) DK_W(5) DK_W(6) DK_W(7) DK_W(8) DK_W(9) DK_W(10) DK_W(11) DK_W(12) DK_W(13) DK_W(14)
    DK_W(15) DK_W(16) DK_W(17) DK_W(18) DK_W(19) DK_W(20) DK_W(21) DK_W(22) DK_W(23) DK_W(24)
#undef DK_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// debug stamps (ca_debug_decode_stamps): thread 0 of a workgroup appends the shader clock at fixed points of the program
struct DkDbg {
  unsigned long long* p;
  int n, i, tid;
};
__device__ __forceinline__ void dk_t(DkDbg& dbg) {
  if (dbg.p && (int)threadIdx.x == dbg.tid && dbg.i < dbg.n) dbg.p[dbg.i++] = __builtin_readcyclecounter();
}

// ---- seams -----------------------------------------------------------------------------------------------------------
// wave 0: wait until every workgroup's progress word is >= target.  Returns false on a timeout (wave-uniform).
__device__ __forceinline__ bool dk_poll(const DecArgs& a, unsigned target, int G, int lane) {
  const unsigned* myflags = a.flags + ((int)blockIdx.x & (a.flag_reps - 1)) * DK_FLAG_STRIDE;
  for (unsigned spins = 0;; ++spins) {
    dk_u32x4 f = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (lane * 4 < G) f = dk_ld16u_sc1(myflags + lane * 4);
    dk_vm0();
    dk_tie(f);
    bool ok = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) ok &= (lane * 4 + e >= G) || f[e] >= target;
    if (__all(ok)) return true;
    if (spins > DK_SPIN_LIMIT) return false;
    switch (a.poll_sleep) {  // (64 clocks per unit; CA_DECODE_POLL_SLEEP)
      case 0: break;
      case 1: __builtin_amdgcn_s_sleep(1); break;
      case 2: __builtin_amdgcn_s_sleep(2); break;
      case 4: __builtin_amdgcn_s_sleep(4); break;
      default: __builtin_amdgcn_s_sleep(8); break;
    }
  }
}
// All waves: the seam in front of phase `ph` (1-based; waits for every workgroup to have completed ph - 1).  Returns
// false when the launch is being abandoned.  `lds_ok`: one LDS word.
__device__ __forceinline__ bool dk_seam(const DecArgs& a, unsigned ph, int G, volatile int* lds_ok, int wave, int lane,
                                        int keep_in_flight, DkDbg& dbg) {
  if (wave == 0) {
    const bool ok = dk_poll(a, ph - 1, G, lane);
    if (lane == 0) {
      *lds_ok = ok ? 1 : 0;
      if (!ok) atomicMax(a.status, ph + 1);
    }
  }
  else {
    // this wave's LDS-DMA pieces - its own and its share of wave 0's - have landed, but for the batch issued last where
    // the coming phase's entry is older than that batch (keep_in_flight = the batch's instructions, else 0)
    dk_wait_vm(keep_in_flight);
  }
  if (a.stamp_fine) dk_t(dbg);
  __syncthreads();
  const bool ok = *lds_ok != 0;
  return ok;
}
// wave 0 (the only wave that stores handed-off bytes): drain, then publish that this workgroup has completed phase ph
__device__ __forceinline__ void dk_publish(const DecArgs& a, unsigned ph, int w, int wave, int lane) {
  if (wave == 0) {
    dk_vm0();
    if (lane < a.flag_reps) dk_st4_sc1(a.flags + lane * DK_FLAG_STRIDE + w, ph);  // ONE wave instruction: every replica's word
  }
}

// ---- columns of a workgroup ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void dk_cols(int N, int w, int G, int& c0, int& nc) {
  const int ng = N >> 2, base = ng / G, rem = ng - base * G;
  const int mine = base + (w < rem ? 1 : 0);
  const int g0 = w * base + (w < rem ? w : rem);
  c0 = 4 * g0;
  nc = 4 * mine;
}

// ---- the weight ring ---------------------------------------------------------------------------------------------------
struct DkEnt {
  const unsigned short* W;  // row c0 of the matrix
  int nc, K, pcs;           // this workgroup's columns, the matrix's K, pieces of a wave's quarter
};
struct DkRing {
  int start, used, next_e, cur_e;
  int batch_first, batch_issued;  // the latest dk_ring_advance: its first entry, the LDS-DMA instructions THIS wave issued
  int next_k;                     // next_e % 6 while next_e is a layer's entry
  int iss_pc, iss_slot0;          // pieces of entry next_e already issued (0: none), its first slot
  unsigned pcs_pack;              // pieces of the six projections, 5 bits each (the bookkeeping of wave 0 sits between its
                                  // publish and its poll: no division, no LDS read)
};
// the six projections of a layer as this workgroup sees them (layer-independent: built once, kept in LDS): first column,
// columns, K, pieces, field of the weight pointer in the layer record; entry 6: K and quads of a vocabulary tile
struct DkKind {
  int c0, nc, K, pcs, fld, nquads;
  int ln_fld, bias_fld;  // fields of the LayerNorm gamma (beta = + 1; -1: no LayerNorm) and of the bias in the layer record
};
__device__ __forceinline__ int dk_per(int K) { return (((K + 31) >> 5) + 3) >> 2; }  // k-steps of a wave's K quarter
template <class T>
__device__ __forceinline__ T dk_lyp(const unsigned long long* ltab, int l, int field) {
  return (T)(uintptr_t)ltab[l * LY_WORDS + field];
}
__device__ __forceinline__ void dk_build_kinds(const DecArgs& a, DkKind* kt, int w, int G) {
  const int t = threadIdx.x;
  if (t < 7) {
    DkKind k;
    int N = a.d;
    k.K = a.d;
    k.ln_fld = -1;
    k.bias_fld = 0;
    switch (t) {
      case 0: k.fld = LY_WQKV; N = 3 * a.d; k.ln_fld = LY_LN1G; k.bias_fld = LY_BQKV; break;
      case 1: k.fld = LY_WO; k.bias_fld = LY_BO; break;
      case 2: k.fld = LY_WQ2; k.ln_fld = LY_LN2G; k.bias_fld = LY_BQ2; break;
      case 3: k.fld = LY_WO2; k.bias_fld = LY_BO2; break;
      case 4: k.fld = LY_W1; N = a.f; k.ln_fld = LY_LN3G; k.bias_fld = LY_B1; break;
      case 5: k.fld = LY_W2; k.K = a.f; k.bias_fld = LY_B2; break;
      default: k.fld = 0; N = 0; break;  // a vocabulary tile: K, quads
    }
    k.c0 = 0;
    k.nc = 0;
    if (t < 6) dk_cols(N, w, G, k.c0, k.nc);
    k.nquads = (dk_per(k.K) + 3) >> 2;
    k.pcs = k.nc > 0 ? ((k.nc + 3) >> 2) * k.nquads : 0;
    kt[t] = k;
  }
}
// pieces of entry e (cheap: what wave 0, which issues nothing, needs)
__device__ __forceinline__ int dk_entry_pcs(const DecArgs& a, const DkKind* kt, int e, int w, int G) {
  const int L6 = 6 * a.n_layers;
  if (e < L6) return kt[e % 6].pcs;
  const int c0 = 16 * (w + G * (e - L6));
  const int nc = a.V - c0 < 16 ? a.V - c0 : 16;
  return nc > 0 ? ((nc + 3) >> 2) * kt[6].nquads : 0;
}
__device__ __forceinline__ DkEnt dk_entry(const DecArgs& a, const unsigned long long* ltab, const DkKind* kt, int e, int w, int G) {
  DkEnt r;
  const int L6 = 6 * a.n_layers;
  if (e < L6) {
    const int l = e / 6, p = e - 6 * l;
    const DkKind k = kt[p];
    r.nc = k.nc;
    r.K = k.K;
    r.pcs = k.pcs;
    r.W = dk_lyp<const unsigned short*>(ltab, l, k.fld) + (int64_t)k.c0 * k.K;
  } else {
    const int c0 = 16 * (w + G * (e - L6));
    r.nc = a.V - c0 < 16 ? a.V - c0 : 16;
    r.K = a.d;
    r.pcs = r.nc > 0 ? ((r.nc + 3) >> 2) * kt[6].nquads : 0;
    r.W = a.embed + (int64_t)c0 * a.d;
  }
  return r;
}
// piece pc of an entry = column group pc / nquads, k-step quad pc % nquads: lane i brings chunk (i & 15) ^ (4 (i >> 4)) of
// column i >> 4 (16-byte chunks of the quad's 128 k) to position i
// Waves 1 .. 3 issue the LDS-DMA: their own pieces and, dealt round-robin, wave 0's - wave 0 polls the seams, and a load
// of its own behind freshly issued pieces would wait for them to land (vmcnt counts in order: a 64 KB refill in front of the
// poll measured +3 .. 4 us on the seam).  The issuing waves wait for their pieces in front of the seam's barrier, where they
// idle anyway - for all but the batch issued last, which no phase needs yet; behind that barrier the pieces of the next
// phase are in LDS for all four waves.
// (no piece needs zeros: columns beyond the workgroup's and k-steps beyond the quarter's are never read back as operands
// that reach a stored output - their lanes fetch a valid neighbour instead.  EXACT: the quarter is whole quads, PD % 4 == 0.)
// Pieces [pc0, pc1) of an entry.
template <bool EXACT>
__device__ __forceinline__ int dk_issue_pieces(const DkEnt& e, int slot0, int pc0, int pc1, uint32_t ring0_lds, int wave, int lane) {
  // wave 1 .. 3: every piece of its own quarter of K, and every third piece (pc % 3 == wave - 1) of wave 0's
  const int per = e.K >> 7, nquads = (per + 3) >> 2;
  const int c = lane >> 4, jj = (lane & 15) ^ (4 * c);
  const int kq = jj >> 2;  // k-step inside a quad
  const int own_k = wave * per * 32;
  const uint32_t own_lds = ring0_lds + (uint32_t)wave * (DK_RP * 1024u);
  int cg = 0, S = pc0;
  while (S >= nquads) {
    S -= nquads;
    ++cg;
  }
  int third = (pc0 + 3 - (wave - 1)) % 3;  // 0 when pc % 3 == wave - 1
  int slot = slot0 + pc0;
  slot = slot >= DK_RP ? slot - DK_RP : slot;
  int n = 0;
  for (int pc = pc0; pc < pc1; ++pc) {
    int col = 4 * cg + c;
    col = col < e.nc ? col : e.nc - 1;
    int ks = 4 * S + kq;
    if (!EXACT) ks = ks < per ? ks : per - 1;
    const unsigned short* src = e.W + (int64_t)col * e.K + 8 * (jj & 3) + ks * 32;
    if (DK_NT & 1) dk_glds16_nt(src + own_k, own_lds + (uint32_t)slot * 1024u); else dk_glds16(src + own_k, own_lds + (uint32_t)slot * 1024u);
    ++n;
    if (third == 0) {
      if (DK_NT & 1) dk_glds16_nt(src, ring0_lds + (uint32_t)slot * 1024u); else dk_glds16(src, ring0_lds + (uint32_t)slot * 1024u);
      ++n;
    }
    third = third == 2 ? 0 : third + 1;
    if (++S == nquads) {
      S = 0;
      ++cg;
    }
    slot = slot + 1 >= DK_RP ? 0 : slot + 1;
  }
  return n;
}
// Refill the ring: at most `budget` pieces per call (an entry may be issued over several calls: an LDS-DMA instruction costs
// its wave ~350 clocks here, and a burst of 16 between a phase's end and the next seam's barrier was on every workgroup's
// critical path), but never less than what completes the entry the NEXT projection consumes.
template <bool EXACT>
__device__ __forceinline__ void dk_ring_advance(const DecArgs& a, const unsigned long long* ltab, const DkKind* kt, DkRing& rg,
                                                int n_entries, int w, int G, uint32_t ring_lds, int wave, int lane, int budget) {
  rg.batch_first = rg.next_e;
  rg.batch_issued = 0;
  const int L6 = 6 * a.n_layers;
  while (rg.next_e < n_entries) {
    int pcs;
    if (rg.next_e < L6) {
      pcs = (int)((rg.pcs_pack >> (5 * rg.next_k)) & 31u);
    } else {
      const int c0 = 16 * (w + G * (rg.next_e - L6));
      const int nc = a.V - c0 < 16 ? a.V - c0 : 16;
      pcs = nc > 0 ? ((nc + 3) >> 2) * (int)(rg.pcs_pack >> 30) : 0;
    }
    const bool urgent = rg.next_e <= rg.cur_e;
    if (rg.iss_pc == 0) {  // a new entry: its whole space is taken now
      if (rg.used + pcs > DK_RP) break;
      if (budget <= 0 && !urgent) break;
      rg.iss_slot0 = rg.start + rg.used;
      rg.iss_slot0 = rg.iss_slot0 >= DK_RP ? rg.iss_slot0 - DK_RP : rg.iss_slot0;
      rg.used += pcs;
    } else if (budget <= 0 && !urgent) {
      break;
    }
    int upto = urgent ? pcs : rg.iss_pc + budget;
    upto = upto < pcs ? upto : pcs;
    if (upto > rg.iss_pc && wave != 0) {
      const DkEnt e = dk_entry(a, ltab, kt, rg.next_e, w, G);
      rg.batch_issued += dk_issue_pieces<EXACT>(e, rg.iss_slot0, rg.iss_pc, upto, ring_lds, wave, lane);
    }
    budget -= upto - rg.iss_pc;
    rg.iss_pc = upto;
    if (rg.iss_pc < pcs) break;  // (budget spent inside the entry)
    rg.iss_pc = 0;
    ++rg.next_e;
    rg.next_k = rg.next_k == 5 ? 0 : rg.next_k + 1;
  }
}
__device__ __forceinline__ void dk_ring_pop(DkRing& rg, int pcs) {
  rg.start += pcs;
  rg.start = rg.start >= DK_RP ? rg.start - DK_RP : rg.start;
  rg.used -= pcs;
  ++rg.cur_e;
}
// MFMA A fragment (weights) of 16-column tile t16, k-step s of the wave's quarter, from the ring entry at `start`: the lane's
// column group and its four chunk offsets (one per k-step of a quad) are fixed per tile
struct DkWf {
  int slot0;   // ring slot of the lane's column group, quad 0
  int off[4];  // byte offset inside a piece for k-step s & 3
};
__device__ __forceinline__ DkWf dk_wfrag_setup(int start, int groups, int nquads, int t16, int lane) {
  const int r = lane & 15, g = lane >> 4;
  int cg = 4 * t16 + (r >> 2);
  cg = cg < groups ? cg : 0;  // (rows beyond the workgroup's columns: a copy, dropped by the epilogue)
  const int c = r & 3;
  DkWf f;
  f.slot0 = start + cg * nquads;
#pragma unroll
  for (int q = 0; q < 4; ++q) f.off[q] = (16 * c + ((4 * q + g) ^ (4 * c))) * 16;
  return f;
}
__device__ __forceinline__ bf16x8_t dk_wfrag(const char* ring, const DkWf& f, int s) {
  int slot = f.slot0 + (s >> 2);
  slot = slot >= DK_RP ? slot - DK_RP : slot;
  return *(const bf16x8_t*)(ring + slot * 1024 + f.off[s & 3]);
}

// wave_sum's butterfly (v += v[lane ^ o] for o = 32, 16, 8, 4, 2, 1 - the same pairs, so the same bits) on four values at
// once with one trip through the LDS crossbar instead of six: ds_bpermute for ^32, v_permlane16_swap for ^16 (of x = y = v
// it leaves the two partners of every lane in x and y: x + y is the step's result, the operand order being immaterial), DPP
// for the rest (^8 = row_ror:8, ^4 = half-row mirror of the quad's mirror, ^2 and ^1 quad permutes).  Six ds_bpermute_b32 and their
// waits per reduction made the LayerNorm of 16 rows 3.7 us of every projection phase that starts with one.
template <int CTRL>
__device__ __forceinline__ float dk_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float dk_bfly_sum(float v) {
#ifdef DK_SHFL_BFLY
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#endif
  // (^32: ds_bpermute.  v_permlane32_swap_b32 on this chip left zeros in the low half of its second operand,
  // tools/r06/bfly_test.hip; and hipcc 7.2 models only the first result of the permlane*_swap builtins, hence inline asm)
  v += __shfl_xor(v, 32, 64);
  {
    unsigned x = __builtin_bit_cast(unsigned, v), y = x;
    // (s_nop: the VALU -> permlane and permlane -> VALU wait states the assembler does not insert for inline asm)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    v = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
  }
  v += dk_dpp<0x128>(v);                 // row_ror:8
  v += dk_dpp<0x141>(dk_dpp<0x1B>(v));   // quad_perm [3,2,1,0], then row_half_mirror: lane ^ 3 ^ 7 = lane ^ 4
  v += dk_dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
  v += dk_dpp<0xB1>(v);                  // quad_perm [1,0,3,2]
  return v;
}
// The LayerNorm arithmetic of ln_fwd_kernel / ca_gemm_skinny_kernel's prologue AS THOSE KERNELS ARE COMPILED: the squared
// deviations are rounded products added one by one (their loops are packed into v_pk_mul_f32 + v_add_f32: no fused
// multiply-add), the affine part is one fused multiply-add on the rounded (x - mean) * rstd.  Written out here - contraction
// off, the one fusion explicit - because the compiler's choice depends on the shape of the code around the expression
// (four rows side by side compiled to other fusions: different bits in 2 rows of 16, tools/r06/persist_check.py).
__device__ __forceinline__ float dk_sq_acc(float s2, float x, float mean) {
#pragma clang fp contract(off)
  const float dlt = x - mean;
  const float p = dlt * dlt;
  return s2 + p;
}
__device__ __forceinline__ float dk_ln_apply(float x, float mean, float rstd, float gm, float bt) {
#pragma clang fp contract(off)
  const float t = (x - mean) * rstd;
  return __builtin_fmaf(t, gm, bt);
}
// ---- LayerNorm of the B rows into the LDS image (the arithmetic of ca_gemm_skinny_kernel's prologue = ln_fwd_kernel) ---------
// EMB: the rows are token + position embeddings formed here (embed_kernel's arithmetic) instead of loaded
// LayerNorm vectors and biases reach their phase through LDS too (LDS-DMA by waves 1 .. 3, behind a ring batch so that the
// seam's counted wait leaves them in flight): gamma | beta of the NEXT LayerNorm phase are asked for when the previous one
// is two seams away and the scratch region is not an attention phase's (after B for D, after E for G, after G for the next
// layer's A or the output projection), the six biases of the next layer (this workgroup's columns) during phase A.  As loads
// in front of the seam they put an HBM round trip in front of wave 0's poll (vmcnt counts in order).
#define DK_PSLOT 49152                 // in the scratch region, above the LayerNorm image: gamma (6 KiB) | beta (6 KiB)
#define DK_BSLOT 12288                 // in the first region: 2 layers x 6 projections x 128 B (32 columns)
__device__ __forceinline__ int dk_issue_ln(const float* gamma, const float* beta, int d, uint32_t pslot_lds, int wave, int lane) {
  if (wave != 1 && wave != 2) return 0;
  const float* src = wave == 1 ? gamma : beta;
  const uint32_t dst = pslot_lds + (wave == 1 ? 0u : 6144u);
  const int nch = d >> 2;  // 16-byte chunks
  int n = 0;
  for (int i = 0; i * 64 < nch; ++i) {
    const int ch = i * 64 + lane;
    dk_glds16(ch < nch ? (const void*)(src + ch * 4) : (const void*)g_dec_zero_page, dst + (uint32_t)i * 1024u);
    ++n;
  }
  return n;
}
__device__ __forceinline__ int dk_issue_biases(const unsigned long long* ltab, const DkKind* kt, int l, uint32_t bslot_lds, int wave,
                                               int lane) {
  if (wave != 3) return 0;
  int n = 0;
  for (int k = 0; k < 6; ++k) {
    const DkKind kk = kt[k];
    const float* bias = dk_lyp<const float*>(ltab, l, kk.bias_fld);
    if (kk.nc <= 0) continue;
    const bool ok = bias != nullptr && lane * 4 < kk.nc;
    if (lane < 8)  // (the inactive lanes write nothing: 128 bytes per slot)
      dk_glds16(ok ? (const void*)(bias + kk.c0 + lane * 4) : (const void*)g_dec_zero_page,
                bslot_lds + (uint32_t)(((l & 1) * 6 + k) * 128));
    ++n;
  }
  return n;
}
// gamma / beta of a LayerNorm into registers: asked for BEFORE the seam in front of the phase (parameters, not handed-off
// bytes), so their trip to HBM runs while the seam resolves.  NCX: d is a multiple of 512 (every lane has NC whole chunks).
template <int NC, bool NCX>
__device__ __forceinline__ void dk_ln_params(const char* pslot, int C, int lane, f32x4_t (&gq)[NC][2], f32x4_t (&bq)[NC][2]) {
  const int nchunk = C >> 3;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + c * 64;
    if (NCX || c < NC - 1 || ch < nchunk) {
      gq[c][0] = *(const f32x4_t*)(pslot + ch * 32);
      gq[c][1] = *(const f32x4_t*)(pslot + ch * 32 + 16);
      bq[c][0] = *(const f32x4_t*)(pslot + 6144 + ch * 32);
      bq[c][1] = *(const f32x4_t*)(pslot + 6144 + ch * 32 + 16);
    } else {
      gq[c][0] = gq[c][1] = bq[c][0] = bq[c][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  }
}
// LayerNorm of the 16 rows into the LDS image (the arithmetic of ca_gemm_skinny_kernel's prologue = ln_fwd_kernel): a wave
// takes rows wave, wave + 4, ...; rows beyond B are formed from zeros and not stored.  emb: the rows are token + position
// embeddings formed here (embed_kernel's arithmetic) instead of loaded.
// NR: rows per wave (4; 2 where B <= 8: rows wave + 8 and wave + 12 do not exist)
template <int NC, bool NCX, int NR>
__device__ __forceinline__ void dk_ln_rows(const DecArgs& a, const unsigned short* x, bool emb, const f32x4_t (&gq)[NC][2],
                                           const f32x4_t (&bq)[NC][2], unsigned short* xs, int wave, int lane) {
  const int C = a.d, nchunk = C >> 3, xpitch = C + DK_XPAD;
  const u16x8_t z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  u16x8_t raw[NR][NC];
  if (!emb) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = wave + 4 * i;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        const bool ok = row < a.B && (NCX || c < NC - 1 || ch < nchunk);
        raw[i][c] = ok ? dk_ld16_sc1(x + (int64_t)row * C + ch * 8) : z8;
      }
    }
    dk_vm0();
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int c = 0; c < NC; ++c) dk_tie(raw[i][c]);
  } else {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = wave + 4 * i;
      int64_t t_off = 0, p_off = 0;
      if (row < a.B) {
        t_off = (int64_t)a.tok[row] * C;
        p_off = (int64_t)a.pos[row] * C;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        const bool ok = row < a.B && (NCX || c < NC - 1 || ch < nchunk);
        const u16x8_t ta = ok ? *(const u16x8_t*)(a.embed + t_off + ch * 8) : z8;
        const u16x8_t pa = ok ? *(const u16x8_t*)(a.pos_tab + p_off + ch * 8) : z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[i][c][e] = f2bf(bf2f(ta[e]) + bf2f(pa[e]));
      }
    }
  }
  // the rows of a wave side by side (independent reduction chains overlap their latencies); per row the arithmetic and its
  // order are ln_fwd_kernel's (dk_sq_acc / dk_ln_apply: with the fusions written out).  The rows as fp32 once, for the
  // three passes.
  float xf[NR][NC][8];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[i][c][e] = bf2f(raw[i][c][e]);
  float sx[NR], mean[NR], s2[NR], rstd[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    sx[i] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (NCX || c < NC - 1 || lane + c * 64 < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sx[i] += xf[i][c][e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) sx[i] = dk_bfly_sum(sx[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    mean[i] = sx[i] / (float)C;
    s2[i] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (NCX || c < NC - 1 || lane + c * 64 < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s2[i] = dk_sq_acc(s2[i], xf[i][c][e], mean[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) s2[i] = dk_bfly_sum(s2[i]);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int lrow = wave + 4 * i;
    rstd[i] = rsqrtf(s2[i] / (float)C + a.eps);
    if (lrow < a.B) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (NCX || c < NC - 1 || ch < nchunk) {
          u16x8_t o8;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            o8[e] = f2bf(dk_ln_apply(xf[i][c][e], mean[i], rstd[i], e < 4 ? gq[c][0][e] : gq[c][1][e - 4],
                                     e < 4 ? bq[c][0][e] : bq[c][1][e - 4]));
          *(u16x8_t*)(xs + (int64_t)lrow * xpitch + ch * 8) = o8;
        }
      }
    }
  }
}

// ---- a projection ------------------------------------------------------------------------------------------------------
// out[m, c0 + j] = epilogue(sum_k A[m, k] W[c0 + j, k] + bias) for this workgroup's columns of one ring entry.  A: the
// LayerNorm image in LDS (ln) or bf16 rows in global memory (handed-off bytes: sc1 loads).  K is a multiple of 128: a wave's
// quarter is `per` = K / 128 whole k-steps (PF = the unroll bound).  The epilogue functor is called by wave 0's lanes with a
// valid row m for every group of four columns j0 .. j0 + 3 they hold: epi(m, j0, v[4]), v = (p0 + p1) + (p2 + p3) + bias -
// the combine of ca_gemm_skinny_kernel.
template <int PD, class Epi>
__device__ __forceinline__ void dk_project(const DecArgs& a, const DkEnt& ent, int start, bool ln, const unsigned short* Ag,
                                           const unsigned short* xs, const char* ring, float* part, const f32x4_t (&bias4)[2],
                                           int wave, int lane, Epi epi) {
  constexpr int PF = 4 * PD;
  const int r = lane & 15, g = lane >> 4;
  const int K = ent.K, nquads = K > 128 * PD ? PD : (PD + 3) >> 2;  // (K = d: PD k-steps per wave; K = f = 4 d: PF)
  const bool wide = K > 128 * PD;
  const int ks0 = wave * (wide ? PF : PD);
  const int groups = (ent.nc + 3) >> 2;
  // the activation fragments of the wave's K quarter, PD k-steps at a time (K = f: four batches, the next one's loads
  // in flight under the current one's MFMAs - loads complete in order, so vmcnt(PD) says the older batch has landed;
  // 2 x PD fragments of registers instead of 4 x PD)
  bf16x8_t af[2][PD];
  const unsigned short* ap = Ag + (int64_t)(r < a.B ? r : a.B - 1) * K + ks0 * 32 + 8 * g;  // (rows beyond B: a copy, dropped)
  const unsigned short* xp = xs + (int64_t)r * (K + DK_XPAD) + ks0 * 32 + 8 * g;
  if (!ln) {
#pragma unroll
    for (int s = 0; s < PD; ++s) af[0][s] = __builtin_bit_cast(bf16x8_t, dk_ld16_sc1(ap + s * 32));
    if (wide) {
#pragma unroll
      for (int s = 0; s < PD; ++s) af[1][s] = __builtin_bit_cast(bf16x8_t, dk_ld16_sc1(ap + (PD + s) * 32));
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD) : "memory");
    } else {
      dk_vm0();  // the activation rows have landed
    }
#pragma unroll
    for (int s = 0; s < PD; ++s) dk_tie(af[0][s]);
  } else {
#pragma unroll
    for (int s = 0; s < PD; ++s) af[0][s] = *(const bf16x8_t*)(xp + s * 32);
  }
  const int ntile = (ent.nc + 15) >> 4;  // (two tiles only where N / CUs > 16: fc1 of whisper-large; K = d there)
  for (int t16 = 0; t16 < ntile; ++t16) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const DkWf wfs = dk_wfrag_setup(start, groups, nquads, t16, lane);
    {
      bf16x8_t wf[PD];  // (all fragment reads of a batch in flight before the first MFMA waits)
#pragma unroll
      for (int s = 0; s < PD; ++s) wf[s] = dk_wfrag(ring, wfs, s);
#pragma unroll
      for (int s = 0; s < PD; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], af[0][s], acc, 0, 0, 0);
    }
    if (wide) {  // (non-LayerNorm, one tile: fc2)
#pragma unroll
      for (int b = 1; b < 4; ++b) {
        bf16x8_t wf[PD];
#pragma unroll
        for (int s = 0; s < PD; ++s) wf[s] = dk_wfrag(ring, wfs, b * PD + s);
        // batch b sits in af[b & 1]; batch b + 1 goes where batch b - 1 was
        if (b < 3) {
#pragma unroll
          for (int s = 0; s < PD; ++s) af[(b + 1) & 1][s] = __builtin_bit_cast(bf16x8_t, dk_ld16_sc1(ap + ((b + 1) * PD + s) * 32));
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD) : "memory");
        } else {
          dk_vm0();
        }
#pragma unroll
        for (int s = 0; s < PD; ++s) dk_tie(af[b & 1][s]);
#pragma unroll
        for (int s = 0; s < PD; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], af[b & 1][s], acc, 0, 0, 0);
      }
    }
    if (t16) __syncthreads();  // (wave 0 has read the previous tile's partials)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave * 256 + r * 16 + 4 * g + e] = acc[e];
    __syncthreads();
    const int j0 = 16 * t16 + 4 * g;
    if (wave == 0 && r < a.B && j0 < ent.nc) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = r * 16 + 4 * g + e;
        v[e] = (part[i] + part[256 + i]) + (part[512 + i] + part[768 + i]);
        v[e] = v[e] * 1.0f + (t16 ? bias4[1][e] : bias4[0][e]);
      }
      epi(r, j0, v);
    }
  }
}
// the epilogue's bias of wave 0's lanes from the projection's LDS slot (at most two 16-column tiles)
__device__ __forceinline__ void dk_bias4(const char* slot, int nc, int wave, int lane, f32x4_t (&bias4)[2]) {
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bias4[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (wave == 0 && slot && 16 * t + 4 * g < nc) bias4[t] = *(const f32x4_t*)(slot + (16 * t + 4 * g) * 4);
  }
}
// four bf16 of consecutive columns: one 8-byte write-through store
__device__ __forceinline__ void dk_store4(unsigned short* dst, const unsigned (&hv)[4]) {
  dk_u32x2 pk = {hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16)};
  dk_st8_sc1(dst, pk);
}

// ---- single-query attention over a K|V cache: the arithmetic of attn_fwd_smallq_kernel<64, false, D> -------------------------
__device__ __forceinline__ int dk_mnswz8(int kr) { return (((kr >> 1) & 1) << 1) | (((kr >> 3) & 1) << 2); }
__device__ __forceinline__ int dk_rowperm(int bb, int r) { return 8 * (r >> 2) + 4 * bb + (r & 3); }
__device__ __forceinline__ bf16x8_t dk_pack8(const float (&v)[8]) {
  dk_s16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(v[e]);
  return __builtin_bit_cast(bf16x8_t, o);
}
__device__ __forceinline__ bf16x8_t dk_tr_pair(const char* a0, const char* a1) {
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(dk_lds_addr(a0)));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(dk_lds_addr(a1)));
  dk_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t dk_vfrag(const char* img, int s, int nb, int lane) {  // timg_frag_async<64>
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int kr = 32 * s + 8 * g + q;
  const int c = ((2 * nb) + (p >> 1)) ^ dk_mnswz8(kr);
  const char* a0 = img + kr * 128 + c * 16 + (p & 1) * 8;
  return dk_tr_pair(a0, a0 + 4 * 128);
}
// One (clip, head[, key split]) item.  Q: the query row (64 bf16, handed-off bytes); K, V: the head's columns of row 0 of
// the clip's cache (row strides ldk = ldv = 2d); Tk: rows of the cache (clamp bound), kl: valid keys; vw / nvw: this
// wave's place among the 4 * ns waves that share the keys.  smem: DK_SCRATCH.  Wave 0 ends with the merged (M, L, o) of
// the four waves; ns == 1: it stores the output row O (64 bf16, sc1); ns > 1: its partial into `slab`.
// fresh_row >= 0 (the self-attention): row fresh_row of K and V was written by other workgroups in this launch - the K
// fragments are then sc1 loads, and V's row, which arrives in its tile by LDS-DMA (not a load the write-through hand-off
// covers), is fetched again by an sc1 register load and written over the tile's copy in LDS: every handed-off byte that
// is used came through an sc1 load to registers, so the phase needs no acquire (1.5 us per layer).
// `hook`: called once the first tiles are in flight and before the query row is read - the seam in front of the phase where
// K and V do not depend on it (the encoder K|V: 128 KB per CU under way while the seam resolves); false = give up.
template <class Hook>
__device__ __forceinline__ bool dk_attend(const unsigned short* Q, const unsigned short* K, const unsigned short* V, int64_t ldkv,
                                          int Tk, int kl, int ns, int sp, float c2, char* smem, unsigned short* O, float* slab,
                                          int wave, int lane, bool pre, int fresh_row, Hook hook) {
  constexpr int D = 2, IMG = 64 * 64 * 2, NKS = 2, NNB = 4;
  const int g = lane >> 4, r = lane & 15;
  const int vw = sp * 4 + wave, nvw = 4 * ns;
  char* Vring = smem + wave * D * IMG;
  bf16x8_t qf[NKS];
  const int ntile = (kl + 63) / 64;
  const int nw = ntile > vw ? (ntile - vw + nvw - 1) / nvw : 0;
  float m = DK_NEG_BIG, l = 0.f;
  f32x4_t o[NNB];
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) o[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t kf[D][4][NKS];
  u16x8_t vfresh = {0, 0, 0, 0, 0, 0, 0, 0};  // lanes 0 .. 7: chunk `lane` of V's fresh row
  auto issue = [&](auto slot_c, int j) {
    constexpr int S = decltype(slot_c)::value;
    const int kt = vw + nvw * j;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      int key = kt * 64 + 32 * (blk >> 1) + dk_rowperm(blk & 1, r);
      key = key < Tk ? key : Tk - 1;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)  // (sc1 in both attention phases: one code path; the self-attention needs it)
        if ((DK_NT & 4) && fresh_row < 0)
          kf[S][blk][ks] = __builtin_bit_cast(bf16x8_t, dk_ld16_nt(K + (int64_t)key * ldkv + 32 * ks + 8 * g));
        else
          kf[S][blk][ks] = __builtin_bit_cast(bf16x8_t, dk_ld16_sc1(K + (int64_t)key * ldkv + 32 * ks + 8 * g));
    }
    const uint32_t img = dk_lds_addr(Vring + S * IMG);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kr = i * 8 + lane / 8;
      const int c = (lane % 8) ^ dk_mnswz8(kr);
      const int row = kt * 64 + kr;
      const void* src = row < Tk ? (const void*)(V + (int64_t)row * ldkv + c * 8) : (const void*)g_dec_zero_page;
      if (DK_NT & 2) dk_glds16_nt(src, img + i * 1024); else dk_glds16(src, img + i * 1024);
    }
  };
  auto step = [&](auto slot_c, int j) {
    constexpr int S = decltype(slot_c)::value;
    const int kt = vw + nvw * j;
    const int rem = nw - 1 - j;
    if (rem >= 1)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* Vimg = Vring + S * IMG;
    if (fresh_row >= 0 && kt == (fresh_row >> 6)) {  // (wave-uniform) the tile's copy of the fresh row: replaced
      const int kr = fresh_row & 63;
      if (lane < 8) *(u16x8_t*)(Vring + S * IMG + kr * 128 + ((lane ^ dk_mnswz8(kr)) * 16)) = vfresh;
    }
    bf16x8_t vf[2][NNB];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) vf[s][nb] = dk_vfrag(Vimg, s, nb, lane);
    f32x4_t sc[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        dk_tie(kf[S][blk][ks]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[S][blk][ks], qf[ks], acc, 0, 0, 0);
      }
      sc[blk] = acc;
    }
    auto rest = [&](auto full_c) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_c)::value;
      if constexpr (!FULL) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = kt * 64 + 32 * (blk >> 1) + 8 * g + 4 * (blk & 1) + e;
            sc[blk][e] = key < kl ? sc[blk][e] : DK_NEG_BIG;
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
          if constexpr (!FULL) pv = sc[blk][e] > 0.5f * DK_NEG_BIG ? pv : 0.f;
          p[4 * blk + e] = pv;
          sum += pv;
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float ps[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ps[e] = p[8 * s + e];
        const bf16x8_t pf = dk_pack8(ps);
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) {
          dk_tie(vf[s][nb]);
          o[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf[s][nb], o[nb], 0, 0, 0);
        }
      }
    };
    if (kt * 64 + 64 <= kl)
      rest(std::true_type{});
    else
      rest(std::false_type{});
    if (j + D < nw) issue(slot_c, j + D);
  };
  if (pre) {
    if (nw > 0) issue(std::integral_constant<int, 0>{}, 0);
    if (nw > 1) issue(std::integral_constant<int, 1>{}, 1);
    if (!hook()) return false;
  } else {
    if (!hook()) return false;
    if (nw > 0) issue(std::integral_constant<int, 0>{}, 0);
    if (nw > 1) issue(std::integral_constant<int, 1>{}, 1);
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) qf[ks] = __builtin_bit_cast(bf16x8_t, dk_ld16_sc1(Q + 32 * ks + 8 * g));
  if (fresh_row >= 0 && lane < 8) vfresh = dk_ld16_sc1(V + (int64_t)fresh_row * ldkv + lane * 8);
  dk_vm0();  // (the query; the first tiles, older, with it)
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) dk_tie(qf[ks]);
  dk_tie(vfresh);
  for (int j = 0; j < nw; j += D) {
    step(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nw) step(std::integral_constant<int, 1>{}, j + 1);
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  __syncthreads();
  float* cm = (float*)smem;  // [4][16]
  float* cl = cm + 64;       // [4][16]
  float* co = cl + 64;       // [4][16][64]
  if (g == 0) {
    cm[wave * 16 + r] = m;
    cl[wave * 16 + r] = l;
  }
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) co[(wave * 16 + 4 * g + e) * 64 + 16 * nb + r] = o[nb][e];
  __syncthreads();
  if (wave == 0 && g == 0) {  // query 0 only: lane r takes the output columns 4 r .. 4 r + 3 (one 8-byte store)
    const int q = 0;
    float M = cm[q];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, cm[w * 16 + q]);
    float L = 0.f, wgt[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      wgt[w] = __builtin_amdgcn_exp2f(cm[w * 16 + q] - M);
      L = fmaf(cl[w * 16 + q], wgt[w], L);
    }
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v[k] = fmaf(co[(w * 16 + q) * 64 + 4 * r + k], wgt[w], v[k]);
    }
    if (ns > 1) {
      if (r == 0) {
        dk_st4_sc1(slab, __builtin_bit_cast(unsigned, M));
        dk_st4_sc1(slab + 1, __builtin_bit_cast(unsigned, L));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) dk_st4_sc1(slab + 2 + 4 * r + k, __builtin_bit_cast(unsigned, v[k]));
    } else {
      const float inv = L > 0.f ? 1.0f / L : 0.f;
      unsigned hv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) hv[k] = f2bf(v[k] * inv);
      dk_store4(O + 4 * r, hv);
    }
  }
  __syncthreads();  // (the merge buffers are the next item's V rings)
  return true;
}
// wave 0 of the workgroup that drew the last ticket of (clip, head): merge the ns partials in slab order (the arithmetic
// of attn_fwd_smallq_kernel's split merge) and store the output row
__device__ __forceinline__ void dk_merge_split(const float* base, int ns, unsigned short* O, int lane) {
  float Mt[CA_ATTN_SPLIT_MAX], Lt[CA_ATTN_SPLIT_MAX], ot[CA_ATTN_SPLIT_MAX];
  unsigned um[CA_ATTN_SPLIT_MAX], ul[CA_ATTN_SPLIT_MAX], uo[CA_ATTN_SPLIT_MAX];
#pragma unroll
  for (int t = 0; t < CA_ATTN_SPLIT_MAX; ++t) {
    const float* row = base + (t < ns ? t : 0) * DK_SPLIT_ROW;
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(um[t]) : "v"(row) : "memory");
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(ul[t]) : "v"(row + 1) : "memory");
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(uo[t]) : "v"(row + 2 + lane) : "memory");
  }
  dk_vm0();
#pragma unroll
  for (int t = 0; t < CA_ATTN_SPLIT_MAX; ++t) {
    asm volatile("" : "+v"(um[t]), "+v"(ul[t]), "+v"(uo[t]));
    Mt[t] = __builtin_bit_cast(float, um[t]);
    Lt[t] = __builtin_bit_cast(float, ul[t]);
    ot[t] = __builtin_bit_cast(float, uo[t]);
  }
  float M = DK_NEG_BIG;
#pragma unroll
  for (int t = 0; t < CA_ATTN_SPLIT_MAX; ++t) M = fmaxf(M, t < ns ? Mt[t] : DK_NEG_BIG);
  float L = 0.f, v = 0.f;
#pragma unroll
  for (int t = 0; t < CA_ATTN_SPLIT_MAX; ++t) {
    const float wt = t < ns ? __builtin_amdgcn_exp2f(Mt[t] - M) : 0.f;
    L = fmaf(Lt[t], wt, L);
    v = fmaf(ot[t], wt, v);
  }
  const float inv = L > 0.f ? 1.0f / L : 0.f;
  const float out = v * inv;
  unsigned hv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) hv[k] = f2bf(__shfl(out, 4 * (lane & 15) + k, 64));
  if (lane < 16) dk_store4(O + 4 * lane, hv);
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------
// One loop over the 8 L + 1 steps of a token with ONE site for the projection code and ONE for the attention code: written as
// eight specialised phases per layer the kernel was 188 KB of instructions - three times the instruction cache, and ~31 k
// instructions per wave and layer at one wave per SIMD (1.87 ms per token at 16 clips, instruction-issue bound).
// PD = d / 128 (k-steps of a wave's quarter of K = d; K = f = 4 d: 4 PD), NC = 64-lane rounds of 8-element chunks in a row.
template <int PD, int NC, bool NCX>
__global__ __launch_bounds__(256) void whisper_decode_token_kernel(const DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) char dk_smem[];
  constexpr int PF = 4 * PD;
  float* part = (float*)dk_smem;
  volatile int* lds_ok = (volatile int*)(dk_smem + 4096);
  DkKind* kt = (DkKind*)(dk_smem + 4096 + 64);  // (7 x 32 bytes behind the seam word)
  unsigned long long* ltab = (unsigned long long*)(dk_smem + 5120);
  char* scratch = dk_smem + DK_PART;
  unsigned short* xs = (unsigned short*)scratch;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* ring = dk_smem + DK_PART + DK_SCRATCH + wave * DK_RP * 1024;
  const uint32_t ring_lds = dk_lds_addr(dk_smem + DK_PART + DK_SCRATCH);  // (wave 0's ring: dk_issue_pieces adds the wave)
  const int G = gridDim.x, w = blockIdx.x;
  const int r = lane & 15, g = lane >> 4;
  const int d = a.d, B = a.B, H = a.H, L = a.n_layers;
  {
    // Every clip had finished before this launch (`done` is only written at the very end of a launch, behind all seams:
    // every workgroup reads the same values here): nothing to decode - the step's bookkeeping alone (a finished row records
    // pad and moves on, ca_argmax_advance; `out`, the logits and the cache keep what they have).  What lets the host keep a
    // chunk of launches queued AHEAD of its all-finished check (WhisperEngine._generate_graph) at no cost.
    bool all = true;
    for (int b = 0; b < B; ++b) all &= a.done[b] != 0;
    if (all) {
      if (w < B && threadIdx.x == 0) {
        const int32_t pz = a.pos[w];
        if (pz + 1 < a.ld_ids) a.ids[(int64_t)w * a.ld_ids + pz + 1] = a.pad;
        a.tok[w] = a.pad;
        a.pos[w] = pz + 1;
        a.klen[w] += 1;
      }
      return;
    }
  }
  const int ntiles = (a.V + 15) >> 4;
  const int my_tiles = ntiles > w ? (ntiles - w + G - 1) / G : 0;
  const int n_entries = 6 * L + (ntiles + G - 1) / G;
  for (int i = threadIdx.x; i < L * LY_WORDS; i += 256) ltab[i] = ((const unsigned long long*)a.layers)[i];
  dk_build_kinds(a, kt, w, G);
  __syncthreads();
  const uint32_t pslot_lds = dk_lds_addr(scratch + DK_PSLOT), bslot_lds = dk_lds_addr(dk_smem + DK_BSLOT);
  const char* pslot = scratch + DK_PSLOT;
  DkRing rg = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
  for (int k = 0; k < 6; ++k) rg.pcs_pack |= (unsigned)kt[k].pcs << (5 * k);
  rg.pcs_pack |= (unsigned)kt[6].nquads << 30;  // (quads of a vocabulary tile: at most 3)
  dk_issue_ln(dk_lyp<const float*>(ltab, 0, LY_LN1G), dk_lyp<const float*>(ltab, 0, LY_LN1B), a.d, pslot_lds, wave, lane);
  dk_issue_biases(ltab, kt, 0, bslot_lds, wave, lane);
  dk_ring_advance<PD % 4 == 0>(a, ltab, kt, rg, n_entries, w, G, ring_lds, wave, lane, 1 << 20);
  unsigned ph = 0;  // phases completed by this workgroup
  DkDbg dbg = {a.stamps ? a.stamps + (int64_t)w * a.stamp_nph : nullptr, a.stamp_nph, 0, a.stamp_tid};
  const int dc0 = kt[1].c0, dnc = kt[1].nc;  // this workgroup's columns of every N = d projection (at most 16)
  // wave 0: the residual stream at (row r, column dc0 + 4 g + e), as stored (bf16).  Layer 0: the embedding rows
  // (embed_kernel's arithmetic)
  float res[4] = {0.f, 0.f, 0.f, 0.f};
  const int mypos = r < B ? a.pos[r] : 0;  // the cache row this token's K|V go to (clip r)
  if (wave == 0 && r < B && 4 * g < dnc) {
    const int64_t t_off = (int64_t)a.tok[r] * d, p_off = (int64_t)mypos * d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = dc0 + 4 * g + e;
      res[e] = bf2f(f2bf(bf2f(a.embed[t_off + n]) + bf2f(a.pos_tab[p_off + n])));
    }
  }
  float best = -__builtin_inff();  // wave 0: the running greedy pick of (row r, this lane's vocabulary columns)
  int bi = 0x7fffffff;
  // the seam in front of the next phase; false = the launch is being abandoned
  auto seam = [&](int keep) {
    const bool ok = dk_seam(a, ph + 1, G, lds_ok, wave, lane, keep, dbg);
    if (ok) dk_t(dbg);
    return ok;
  };
  // `step` is over: publish, refill the ring, and ask for what a later phase reads from the LDS slots (behind the ring
  // batch: the next seam's counted wait leaves both in flight)
  auto done = [&](int step) {
    dk_t(dbg);
    ++ph;
    dk_publish(a, ph, w, wave, lane);
    if (a.stamp_fine) dk_t(dbg);
    dk_ring_advance<PD % 4 == 0>(a, ltab, kt, rg, n_entries, w, G, ring_lds, wave, lane, step < 8 * L - 1 ? a.ring_budget : 1 << 20);
    if (a.stamp_fine) dk_t(dbg);
    const int l = step >> 3, p = step & 7;
    if (step < 8 * L) {
      if (p == 1)
        rg.batch_issued += dk_issue_ln(dk_lyp<const float*>(ltab, l, LY_LN2G), dk_lyp<const float*>(ltab, l, LY_LN2B), a.d, pslot_lds, wave, lane);
      else if (p == 4)
        rg.batch_issued += dk_issue_ln(dk_lyp<const float*>(ltab, l, LY_LN3G), dk_lyp<const float*>(ltab, l, LY_LN3B), a.d, pslot_lds, wave, lane);
      else if (p == 6) {
        if (l + 1 < L)
          rg.batch_issued += dk_issue_ln(dk_lyp<const float*>(ltab, l + 1, LY_LN1G), dk_lyp<const float*>(ltab, l + 1, LY_LN1B), a.d, pslot_lds, wave, lane);
        else
          rg.batch_issued += dk_issue_ln(a.lnf_g, a.lnf_b, a.d, pslot_lds, wave, lane);
      } else if (p == 0 && l + 1 < L)
        rg.batch_issued += dk_issue_biases(ltab, kt, l + 1, bslot_lds, wave, lane);
    }
  };

  const int nsteps = 8 * L + 1;
  for (int step = 0; step < nsteps; ++step) {
    const int l = step >> 3, p = step & 7;
    const bool head = step == 8 * L;
    if (!head && (p == 1 || p == 4)) {
      // ---- attention: B (self, over this layer's cache) or E (over the cached encoder K|V) ----------------------------------------
      const bool cross = p == 4;
      const int nit = cross ? ca_key_split_parts(a.split, B * H) : B * H;
      if (w >= nit && wave != 0) dk_wait_vm(rg.cur_e < rg.batch_first ? rg.batch_issued : 0);
      if (w < nit) {
        const unsigned short* kvbase = dk_lyp<const unsigned short*>(ltab, l, cross ? LY_CROSSKV : LY_SELFKV);
        const int Tk = cross ? a.Te : a.Lmax;
        const unsigned short* Qb = cross ? a.q2 : a.q;
        unsigned short* Ob = cross ? a.ctx2 : a.ctx;
        if (!cross) {
          if (!seam(0)) return;
          if (a.b_fence) {  // (CA_DECODE_B_FENCE=1: an agent acquire in front of the phase, as well)
            if (threadIdx.x == 0) {
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
              dk_vm0();
            }
            __syncthreads();
          }
        }
        for (int it = w; it < nit; it += G) {
          int bh = it, sp = 0, ns = 1, part0 = it;
          if (cross) ca_key_split_item(a.split, it, bh, sp, ns, part0);
          const int b = bh / H, h = bh - b * H;
          int kl = Tk;
          if (!cross) {
            kl = a.klen[b];
            kl = kl < Tk ? kl : Tk;
          }
          // K | V of (clip, head): columns h * 64 .. of the [Tk, 2 d] rows (K, then V at + d) - or, the encoder K|V in
          // the head-major copy (CaDecodeDesc.cross_head_major: [B][2][H][Te][64]), two contiguous 188 KB strips
          const bool hm = cross && a.cross_hm;
          const unsigned short* Kp = hm ? kvbase + ((int64_t)(b * 2) * H + h) * Tk * 64 : kvbase + (int64_t)b * Tk * 2 * d + h * 64;
          const unsigned short* Vp = hm ? Kp + (int64_t)H * Tk * 64 : Kp + d;
          const int ldkv = hm ? 64 : 2 * d;
          unsigned short* O = Ob + (int64_t)b * d + h * 64;
          const bool first_cross = cross && it == w;  // (E: the first tiles are asked for BEFORE the seam: K and V do not
                                                     // depend on this token)
          const bool ok = dk_attend(Qb + (int64_t)b * d + h * 64, Kp, Vp, ldkv, Tk, kl, ns, sp, a.scale * DK_LOG2E, scratch, O,
                                    a.slab + (int64_t)it * DK_SPLIT_ROW, wave, lane, first_cross && a.pre_issue != 0,
                                    cross ? -1 : kl - 1, [&] { return first_cross ? seam(0) : true; });
          if (!ok) return;
          if (ns > 1 && wave == 0) {
            dk_vm0();  // this workgroup's partial is out
            unsigned ticket = 0;
            if (lane == 0)
              ticket = __hip_atomic_fetch_add(a.split_cnt + (int64_t)l * 16 * H + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == (unsigned)(ns - 1)) dk_merge_split(a.slab + (int64_t)part0 * DK_SPLIT_ROW, ns, O, lane);
          }
        }
      }
      done(step);
      continue;
    }
    // ---- a projection: A (LayerNorm + q|k|v), C / F (out-projection + residual), D (LayerNorm + cross query), G (LayerNorm +
    // fc1 + GELU), H (fc2 + residual) or the tied output projection (head: LayerNorm, then this workgroup's vocabulary tiles) ----
    const int kind = head ? 6 : (p == 0 ? 0 : p == 2 ? 1 : p == 3 ? 2 : p == 5 ? 3 : p == 6 ? 4 : 5);
    const DkKind kk = kt[kind];
    const bool ln = kind == 0 || kind == 2 || kind == 4 || kind == 6;
    const int my_nc = head ? (my_tiles > 0 ? 16 : 0) : kk.nc;
    f32x4_t bias4[2];
    const int keep = rg.cur_e < rg.batch_first ? rg.batch_issued : 0;
    if (my_nc > 0) {
      if (step > 0) {
        if (!seam(keep)) return;
      } else {
        if (wave != 0) dk_vm0();  // (the first ring pieces and LDS slots: see dk_issue_pieces)
        __syncthreads();
      }
      if (ln) {
        f32x4_t gq[NC][2], bq[NC][2];
        dk_ln_params<NC, NCX>(pslot, d, lane, gq, bq);
        const unsigned short* x = kind == 2 ? a.h1 : kind == 4 ? a.h2 : a.h;
        if (B > 8)
          dk_ln_rows<NC, NCX, 4>(a, x, step == 0, gq, bq, xs, wave, lane);
        else
          dk_ln_rows<NC, NCX, 2>(a, x, step == 0, gq, bq, xs, wave, lane);
        if (!head) __syncthreads();
      }
      dk_t(dbg);
    } else if (wave != 0) {
      dk_wait_vm(keep);  // (no seam here: the issuing waves still keep the order of what has landed)
    }
    dk_bias4(head ? nullptr : dk_smem + DK_BSLOT + ((l & 1) * 6 + kind) * 128, my_nc, wave, lane, bias4);
    unsigned short* ckv = dk_lyp<unsigned short*>(ltab, head ? 0 : l, LY_SELFKV);
    const int nent = head ? my_tiles : 1;
    for (int t = 0; t < nent; ++t) {
      const DkEnt ent = dk_entry(a, ltab, kt, rg.cur_e, w, G);
      if (head) {
        // this tile's pieces have landed (the issuing waves wait; the next tiles' pieces, issued behind them, stay in
        // flight); the barrier also says that wave 0 has read the previous tile's partials and that the image is written
        if (wave != 0) dk_wait_vm(t == 0 ? 0 : rg.used - ent.pcs);
        __syncthreads();
      }
      if (ent.nc > 0) {
        const unsigned short* Ag = kind == 1 ? a.ctx : kind == 3 ? a.ctx2 : a.gbuf;
        const int n0 = head ? 16 * (w + G * t) : kk.c0;
        dk_project<PD>(a, ent, rg.start, ln, Ag, xs, ring, part, bias4, wave, lane, [&](int m, int j0, const float(&v)[4]) {
          const int n = n0 + j0;
          unsigned hv[4];
          if (kind == 6) {  // logits (fp32) + this lane's running greedy pick (candidates come in increasing index order)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (n + e < a.V) {
                a.logits[(int64_t)m * a.ld_logits + n + e] = v[e];
                if (!(a.suppress && a.suppress[n + e]) && v[e] > best) {
                  best = v[e];
                  bi = n + e;
                }
              }
            }
            return;
          }
          if (kind == 1 || kind == 3 || kind == 5) {  // + residual, rounded as the launch sequence stores it; the next residual
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              hv[e] = f2bf(v[e] + res[e]);
              res[e] = bf2f((unsigned short)hv[e]);
            }
            dk_store4((kind == 1 ? a.h1 : kind == 3 ? a.h2 : a.h) + (int64_t)m * d + n, hv);
            return;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = f2bf(kind == 4 ? gelu_erf(v[e]) : v[e]);
          unsigned short* dst;
          if (kind == 0)  // q | this token's K|V row of the cache (a group of four never straddles d: multiples of 4)
            dst = n < d ? a.q + (int64_t)m * d + n : ckv + ((int64_t)m * a.Lmax + mypos) * 2 * d + (n - d);
          else if (kind == 2)
            dst = a.q2 + (int64_t)m * d + n;
          else
            dst = a.gbuf + (int64_t)m * a.f + n;
          dk_store4(dst, hv);
        });
      }
      dk_ring_pop(rg, ent.pcs);
      if (head) dk_ring_advance<PD % 4 == 0>(a, ltab, kt, rg, n_entries, w, G, ring_lds, wave, lane, 1 << 20);
    }
    if (head) {
      // this workgroup's best per row: lanes (r, g) of wave 0 hold row r
      if (wave == 0) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ob = __shfl_xor(best, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
          }
        }
        if (g == 0) {
          dk_st4_sc1(a.amax_val + w * 16 + r, __builtin_bit_cast(unsigned, best));
          dk_st4_sc1(a.amax_idx + w * 16 + r, (unsigned)bi);
        }
      }
    }
    done(step);
  }
  // ---- the greedy pick of row w and the step's bookkeeping (ca_argmax_advance's) -------------------------------------------------
  if (w < B) {
    if (!seam(0)) return;
    if (wave == 0) {
      float vb = -__builtin_inff();
      int vi = 0x7fffffff;
      unsigned uv[4], ui[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int src = lane * 4 + k;
        uv[k] = 0xff800000u;  // -inf
        ui[k] = 0x7fffffffu;
        if (src < G) {
          asm volatile("global_load_dword %0, %1, off sc1" : "=v"(uv[k]) : "v"(a.amax_val + src * 16 + w) : "memory");
          asm volatile("global_load_dword %0, %1, off sc1" : "=v"(ui[k]) : "v"(a.amax_idx + src * 16 + w) : "memory");
        }
      }
      dk_vm0();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("" : "+v"(uv[k]), "+v"(ui[k]));
        const float ob = __builtin_bit_cast(float, uv[k]);
        const int oi = (int)ui[k];
        if (ob > vb || (ob == vb && oi < vi)) {
          vb = ob;
          vi = oi;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(vb, o, 64);
        const int oi = __shfl_xor(vi, o, 64);
        if (ob > vb || (ob == vb && oi < vi)) {
          vb = ob;
          vi = oi;
        }
      }
      if (lane == 0) {
        vi = vi == 0x7fffffff ? 0 : vi;
        a.out[w] = vi;
        const int32_t pz = a.pos[w];
        const int32_t tk = a.done[w] ? a.pad : vi;
        a.ids[(int64_t)w * a.ld_ids + pz + 1] = tk;
        if (tk == a.eos) a.done[w] = 1;
        a.tok[w] = tk;
        a.pos[w] = pz + 1;
        a.klen[w] += 1;
      }
    }
  }
}

// ---- C ABI --------------------------------------------------------------------------------------------------------------------
static int dk_device_cus() {
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  return ncu;
}
static unsigned long long* g_dk_stamps = nullptr;
static int g_dk_stamp_nph = 0;
// debug: launches from now on leave per-workgroup phase stamps (shader clock) in `device_buf` ([CUs][nph][2] 64-bit words:
// seam passed / phase done); NULL switches them off
extern "C" int ca_debug_decode_stamps(void* device_buf, int32_t nph) {
  g_dk_stamps = (unsigned long long*)device_buf;
  g_dk_stamp_nph = device_buf ? nph : 0;
  return CA_OK;
}
extern "C" int ca_whisper_decode_token_supported(int32_t B, int32_t d, int32_t f, int32_t H, int32_t V) {
  const int G = dk_device_cus();
  if (G < 16 || G > 256) return 0;
  if (B < 1 || B > CA_DECODE_MAX_B) return 0;
  // the Whisper family: d_model 384 / 512 / 768 / 1024 / 1280 (a wave's quarter of K is whole k-steps), f = 4 d, head_dim 64
  if (!(d == 384 || d == 512 || d == 768 || d == 1024 || d == 1280) || H * 64 != d || f != 4 * d) return 0;
  if (d / 4 > 16 * G) return 0;  // at most 16 columns of an N = d projection per workgroup
  if (((int64_t)V + 15) / 16 > (int64_t)DK_MAXTILES * 16 * G) return 0;
  // the widest ring entry must fit the ring (and fc1's columns two MFMA tiles)
  auto pieces = [&](int N, int K) {
    const int ng = N / 4, mine = (ng + G - 1) / G;
    const int per = K / 128;
    return mine * ((per + 3) / 4);
  };
  if (pieces(3 * d, d) > DK_RP || pieces(f, d) > DK_RP || pieces(d, f) > DK_RP || 4 * ((d / 128 + 3) / 4) > DK_RP) return 0;
  if ((f / 4 + G - 1) / G * 4 > 32 || (3 * d / 4 + G - 1) / G * 4 > 32) return 0;
  return 1;
}

extern "C" int ca_whisper_decode_token(const CaDecodeDesc* desc, void* stream) {
  CA_CHECK_ARG(desc && desc->layers && desc->embed && desc->embed_pos && desc->lnf_g && desc->lnf_b && desc->logits &&
                   desc->out && desc->done && desc->ids && desc->tok && desc->pos && desc->klen && desc->ws && desc->status,
               "ca_whisper_decode_token: null pointer");
  const CaDecodeDesc& c = *desc;
  CA_CHECK_ARG(c.n_layers > 0 && c.n_layers <= DK_MAXLAYERS && ca_whisper_decode_token_supported(c.B, c.d, c.f, c.H, c.V),
               "ca_whisper_decode_token: shape not supported (B <= %d, head_dim 64, d_model <= 1536, <= 256 CUs)", CA_DECODE_MAX_B);
  CA_CHECK_ARG(c.Te > 0 && c.max_len > 0 && c.ld_logits >= c.V && c.ld_ids >= c.max_len, "ca_whisper_decode_token: bad sizes");
  CA_CHECK_ARG(c.ws_bytes >= CA_DECODE_WS_BYTES(c.B, c.d, c.f, c.H, c.n_layers) && ((uintptr_t)c.ws % 16) == 0,
               "ca_whisper_decode_token: ws needs CA_DECODE_WS_BYTES bytes, 16-byte aligned");
  const int G = dk_device_cus();
  DecArgs a;
  a.layers = c.layers; a.n_layers = c.n_layers; a.B = c.B; a.d = c.d; a.f = c.f; a.H = c.H; a.Te = c.Te; a.Lmax = c.max_len;
  a.V = c.V;
  // key split of the cross-attention: ca_attn_fwd's rule (smallq_split), so the merge order - and the bits - are its
  a.split.ns = 1; a.split.tail_start = c.B * c.H; a.split.ns_tail = 1;
  if (c.Te >= 1024) {
    static const int cap = [] { const char* e = getenv("CA_ATTN_SPLIT"); return e ? atoi(e) : CA_ATTN_SPLIT_MAX; }();
    a.split = ca_attn_key_split(c.B * c.H, G, cap);
  }
  a.embed = (const unsigned short*)c.embed; a.pos_tab = (const unsigned short*)c.embed_pos;
  a.lnf_g = c.lnf_g; a.lnf_b = c.lnf_b; a.eps = c.eps; a.scale = 0.125f;  // head_dim 64
  a.logits = c.logits; a.ld_logits = c.ld_logits; a.suppress = c.suppress; a.out = c.out; a.done = c.done; a.ids = c.ids;
  a.ld_ids = c.ld_ids; a.tok = c.tok; a.pos = c.pos; a.klen = c.klen; a.pad = c.pad_id; a.eos = c.eos_id;
  char* p = (char*)c.ws;
  a.flags = (unsigned*)p;
  // The progress words in R replicas, each polled by 1 / R of the workgroups (tools/r06/seam_bench.hip: a bare seam 3.25 us
  // with one copy that all 256 workgroups poll, 2.44 at R = 4, 2.55 at R = 8; with a 16 KB gather behind it 5.15 -> 3.49 at
  // R = 8): a publishing workgroup stores all R words with one wave instruction.  CA_DECODE_FLAG_REPLICAS (1 .. 8).
  static const int reps = [] {
    const char* e = getenv("CA_DECODE_FLAG_REPLICAS");
    const int r = e ? atoi(e) : 8;
    return r >= 8 ? 8 : (r >= 4 ? 4 : (r >= 2 ? 2 : 1));
  }();
  a.flag_reps = reps;
  static const int psleep = [] { const char* e = getenv("CA_DECODE_POLL_SLEEP"); return e ? atoi(e) : 2; }();
  a.poll_sleep = psleep;
  constexpr size_t FLAG_BYTES = (size_t)DK_FLAG_REPS_MAX * DK_FLAG_STRIDE * 4;  // 8 KiB
  a.split_cnt = (unsigned*)(p + FLAG_BYTES);
  const size_t zero_bytes = FLAG_BYTES + (size_t)c.n_layers * 16 * c.H * 4;
  p += zero_bytes;
  const size_t row = (size_t)16 * c.d * 2;
  a.q = (unsigned short*)p; p += row;
  a.ctx = (unsigned short*)p; p += row;
  a.h1 = (unsigned short*)p; p += row;
  a.q2 = (unsigned short*)p; p += row;
  a.ctx2 = (unsigned short*)p; p += row;
  a.h2 = (unsigned short*)p; p += row;
  a.h = (unsigned short*)p; p += row;
  a.gbuf = (unsigned short*)p; p += (size_t)16 * c.f * 2;
  a.slab = (float*)p; p += (size_t)16 * c.H * CA_ATTN_SPLIT_MAX * DK_SPLIT_ROW * 4;
  a.amax_val = (float*)p; p += 256 * 16 * 4;
  a.amax_idx = (int*)p; p += 256 * 16 * 4;
  a.status = c.status;
  a.stamps = g_dk_stamps;
  a.stamp_nph = g_dk_stamp_nph;
  { const char* e = getenv("CA_DECODE_STAMP_TID"); a.stamp_tid = e ? atoi(e) : 0; }
  { const char* e = getenv("CA_DECODE_STAMP_FINE"); a.stamp_fine = e ? atoi(e) : 0; }
  static const int pre = [] { const char* e = getenv("CA_DECODE_PREISSUE"); return e ? atoi(e) : 1; }();
  a.pre_issue = pre;
  static const int budget = [] { const char* e = getenv("CA_DECODE_RING_BUDGET"); return e ? atoi(e) : 5; }();
  a.ring_budget = budget > 0 ? budget : 1 << 20;
  a.cross_hm = c.cross_head_major;
  { const char* e = getenv("CA_DECODE_B_FENCE"); a.b_fence = e ? atoi(e) : 0; }
  CA_CHECK_ARG((size_t)(p - (char*)c.ws) <= (size_t)c.ws_bytes, "ca_whisper_decode_token: workspace layout exceeds ws_bytes");
  hipStream_t s = (hipStream_t)stream;
#define DK_KERNELS(X) X(3, 1, false) X(4, 1, true) X(6, 2, false) X(8, 2, true) X(10, 3, false)
  static bool attr = false;
  if (!attr) {
#define DK_ATTR(PD, NC, NCX) \
  hipFuncSetAttribute((const void*)whisper_decode_token_kernel<PD, NC, NCX>, hipFuncAttributeMaxDynamicSharedMemorySize, DK_LDS);
    DK_KERNELS(DK_ATTR)
#undef DK_ATTR
    attr = true;
  }
  // every polled word starts at zero (a memset node when the step is captured in a graph)
  if (hipMemsetAsync(c.ws, 0, zero_bytes, s) != hipSuccess) {
    ca_set_error("ca_whisper_decode_token: hipMemsetAsync failed");
    return CA_ERR_LAUNCH;
  }
#define DK_LAUNCH(PD, NC, NCX) \
  if (c.d == 128 * PD) hipLaunchKernelGGL((whisper_decode_token_kernel<PD, NC, NCX>), dim3(G), dim3(256), DK_LDS, s, a);
  DK_KERNELS(DK_LAUNCH)
#undef DK_LAUNCH
#undef DK_KERNELS
  CA_CHECK_LAUNCH("ca_whisper_decode_token");
  return CA_OK;
}
