// Input side of the path: JPEG decode on the device.
//   mx.image.imread(path, 1)  (reference dataset.py:204,216; OpenCV imdecode -> libjpeg, default parameters:
//   JDCT_ISLOW, fancy upsampling)  ->  HWC uint8 RGB
// for a batch of baseline (SOF0/SOF1, 8-bit, Huffman, one interleaved scan) files of one geometry - the frames of a
// video.  The host only walks the marker segments (a few hundred bytes per file) and copies the entropy-coded
// bytes into one pinned buffer; everything after that runs on the GPU:
//
//   1. Huffman decoding, parallel INSIDE each file.  A Huffman stream has no entry points, but it is
//      self-synchronising: a decoder started at a wrong bit soon falls into step with the true one.  The scan is cut
//      into 256-byte subsequences, one thread each (Klein & Wiseman 2003; Weissenberger & Schmidt 2018/2021):
//        sync pass     every thread decodes its subsequence from its first byte with a guessed state (block start)
//                      and records where - bit position, MCU slot, coefficient index - its last symbol ended;
//        fix-up passes every thread restarts from its predecessor's recorded end; repeated until no record changes.
//                      The first thread of a segment starts from the true state, so a fixed point IS the sequential
//                      decode (induction over the subsequences); typically 2-3 passes;
//        block scan    exclusive prefix sum of the blocks each subsequence completes -> its first block index;
//        write pass    one more decode per subsequence that stores the coefficients (de-zigzagged int16, DC as the
//                      coded difference) into the dense block array;
//        DC scan       segmented prefix sum of the DC differences per component (reset at restart intervals).
//      Byte stuffing (FF 00) is skipped inside the bit reader; RSTn markers split a file into segments whose block
//      ranges are known in advance (found on the host only when the file declares a restart interval).
//   2. Dequantisation + 8x8 inverse DCT, libjpeg's jidctint.c::jpeg_idct_islow arithmetic (13-bit constants, two
//      passes, its range-limit table), one thread per block.
//   3. Chroma upsampling (jdsample.c h2v1 / h2v2 "fancy" triangle filters, box replication when the subsampled plane
//      is at most 2 samples wide, as jinit_upsampler chooses) fused with jdcolor.c's fixed-point YCbCr -> RGB.
// Bit-exact against Pillow / libjpeg-turbo (tests/test_gpu_jpeg.py) and against oracle/jpeg_np.py.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

#ifndef TN_JPEG_SUBSEQ
#define TN_JPEG_SUBSEQ 256
#endif
#ifndef TN_JPEG_RUNIN
#define TN_JPEG_RUNIN 0      // sync pass 0 with a run-in of this many subsequences (measured round 4: 0 / 1 / 2 -> 8.35 / 8.47 / 8.55 ms per batch)
#endif
constexpr int SUBSEQ = TN_JPEG_SUBSEQ;     // bytes of entropy-coded data per decoding thread
constexpr int MAX_SLOTS = 10;   // blocks per MCU (T.81 B.2.3: sum of Hi x Vi <= 10)
constexpr int LUT_SIZE = 65536; // 16-bit prefix -> lut_entry (code length, value size, zig-zag advance), 0 = no such code
constexpr int FAST_BITS = 10, FAST_SIZE = 1 << FAST_BITS;   // first-level table: codes of up to 10 bits (the rest: 0 -> full table)
constexpr int MAX_SYNC = 4096;  // fix-up passes before the stream is declared corrupt


struct FrameDev {               // per file
  uint32_t scan_off, scan_len;  // its entropy-coded bytes in the batch buffer
  uint32_t lut;                 // table set: 8 LUTs (DC ids 0-3, AC ids 0-3) of LUT_SIZE entries
  uint16_t q[3][64];            // quantisation tables of its components, natural order
  uint8_t slot_dc[MAX_SLOTS], slot_ac[MAX_SLOTS];
};

struct Geom {                   // common to all files of a call
  int W, H, ncomp, hmax, vmax, mcux, mcuy, bpm, blocks_per_frame, ri;
  int hs[3], vs[3];
  int slot_comp[MAX_SLOTS], slot_by[MAX_SLOTS], slot_bx[MAX_SLOTS];
  int plane_w[3], plane_h[3], cw[3], ch[3];
  long plane_off[3], planes_per_frame;
};

struct Seg {                    // a stretch of one file's scan between restart markers (the whole scan without them)
  uint32_t frame, byte_start, byte_end, block_base, nblocks, first_sub, nsub;
};

// ------------------------------------------------------------------------------------------------ table entries
// One decoding-table entry (first level, full table): bits 0-4 code length + SSSS = the bits the symbol takes (0: no such code), 5-8 SSSS = the bits of the value
// behind the code, 9-15 how far the zig-zag index moves: 1 for a DC difference, RRRR + 1 for an AC coefficient, 16 for ZRL,
// 64 (= the block ends) for EOB and the undefined run / size pairs.  The symbol loop then needs no DC / AC case.
__host__ __device__ constexpr uint16_t lut_entry(int len, int size, int advance) { return (uint16_t)((len + size) | (size << 5) | (advance << 9)); }
// ------------------------------------------------------------------------------------------------ bit reader
struct Reader {
  const uint8_t *s;   // the file's scan bytes
  uint32_t end;       // end of the segment (positions are relative to s)
  __device__ __forceinline__ uint32_t ld(uint32_t p) const { return p < end ? s[p] : 0u; }
  // index of the data byte that follows / precedes byte p (a stuffed 00 behind an FF is not data)
  __device__ __forceinline__ uint32_t next(uint32_t p) const { return p + 1 + ((ld(p) == 0xFFu && ld(p + 1) == 0u) ? 1u : 0u); }
  __device__ __forceinline__ uint32_t prev(uint32_t p) const { return (p >= 2 && ld(p - 1) == 0u && ld(p - 2) == 0xFFu) ? p - 2 : p - 1; }
};

// where a subsequence's last symbol ended (pos = byte * 8 + bit, state = MCU slot * 64 + coefficient index), the blocks it
// completed, and the start (in_pos, in_state) that record was decoded from
struct SubRec { uint32_t pos; uint16_t state; uint16_t in_state; uint32_t nblk; uint32_t in_pos; };

// Decodes the symbols that START inside [cp .. sub_end) of a segment.  WRITE: stores coefficients of blocks
// first_block.. (stops after max_blocks); otherwise only tracks the state.  Returns the end record.
// The stream is read through a 64-bit buffer (first unread bit at bit 63) filled four bytes at a time when none of
// them is FF; pn is the next byte to load, nb the unread bits buffered.  The exact position of the next symbol - the
// data byte that holds its first bit - is only worked out (walking back over the buffered bytes) once pn has passed
// the end of the subsequence.
// UNST: the segment's bytes have been through jpeg_unstuff_kernel - no FF 00 pairs left, the position of a bit is arithmetic and a
// refill is one unaligned dword (the stuffed form's reader takes a byte-wise path whenever ONE lane of the wave has an FF among its
// next four bytes, and keeps a mask of the buffered stuffed bytes to know where a symbol starts).
template <bool WRITE, bool UNST>
__device__ __forceinline__ SubRec decode_sub(const Reader &rd, uint32_t cp, uint32_t cb, uint32_t sub_end, int slot, int k,
                                             const Geom &g, const FrameDev &f, const uint16_t *__restrict__ luts,
                                             const uint16_t *__restrict__ fast, const uint16_t *sh_fast, bool in_lds,
                                             int16_t *__restrict__ coef, uint32_t first_block,
                                             uint32_t max_blocks, int *__restrict__ err, uint32_t *blk = nullptr) {
  // WRITE: a block that BEGINS in this subsequence is put together in the thread's 128 bytes of LDS (blk; 16-byte groups
  // swizzled by the lane so that the flush does not hit one bank) and leaves as eight 16-byte stores when it ends; only the
  // parts of the blocks that straddle a subsequence boundary go out as single 2-byte stores into the zeroed array.  Storing
  // every coefficient on its own cost 1.34 ms of the write pass's 2.15 ms per 256 720p frames.
  const uint32_t sw = (threadIdx.x >> 1) & 7u;
  bool own = false;
  uint32_t nblk = 0;
  // the Huffman tables of the MCU's slots, 4 bits each (DC id | AC id << 2), in one register pair: picking the symbol's table
  // is then arithmetic instead of a dependent load from the frame record in every symbol's chain
  uint64_t tabs = 0;
  for (int i = 0; i < g.bpm; ++i) tabs |= (uint64_t)((f.slot_dc[i] & 3) | ((f.slot_ac[i] & 3) << 2)) << (4 * i);
  // rot: the slots' table nibbles (DC id | AC id << 2) ROTATED so that the current slot's sits in bits 0-3 (the accepted chroma layouts
  // have at most 8 blocks per MCU: 32 bits) - it turns by one nibble at a block end, and the symbol loop carries neither the slot
  // number nor a shift by it (the slot of the end record is (first slot + blocks completed) mod blocks per MCU)
  const int slot0 = slot;
  uint32_t tbl;
  const uint32_t rot_sh = 4u * (uint32_t)(g.bpm - 1);
  uint32_t rot = 0;
  for (int i = 0; i < g.bpm; ++i) {
    const int sl = slot + i < g.bpm ? slot + i : slot + i - g.bpm;
    rot |= (uint32_t)((tabs >> (4 * sl)) & 15u) << (4 * i);
  }
  tbl = k == 0 ? (rot & 3u) : (4u | ((rot >> 2) & 3u));
  const uint16_t *lut_base = luts + (size_t)f.lut * 8 * LUT_SIZE;
  // this file's first-level tables: the workgroup's LDS copy (in_lds: a real ds_read - through one generic pointer the lookup
  // was a FLAT load that waits for every outstanding global load as well) or, for a thread of another table set, global memory
  const uint32_t stop = sub_end < rd.end ? sub_end : rd.end;
  uint64_t buf = 0;
  int nb = 0;
  uint32_t pn = cp;
  // pb: the data byte that holds the first unread bit; sm: bit i set = the i-th buffered byte (from the top) is an FF with a
  // stuffed 00 behind it.  Kept up to date as bytes enter and leave the buffer, so that the position of the next symbol is
  // arithmetic (it used to be worked out by walking back over the buffered bytes - two byte loads and a branch per byte, in
  // every symbol step of every lane whose reader had run past the end of its subsequence).
  uint32_t pb = cp, sm = 0;
  // nx: the four bytes at pn, requested when pn last moved - a refill then appends from a register and only ASKS for the next
  // dword; the dependent load (and its wait, which stalls the whole wave) is out of the symbols' chain
  uint32_t nx = 0;
  // UNST: the dword is requested unconditionally (a file's bytes are followed by 16 bytes of padding and the reader never gets
  // further than 12 bytes past a segment's end) and nm masks what lies behind the end to zero when the dword is USED - a branch
  // around the load, or a mask applied to it here, makes hipcc wait for it on the spot.
  uint32_t nm = 0;
  auto fetch = [&]() {
    if constexpr (UNST) {
      const int valid = (int)rd.end - (int)pn;
      nm = valid >= 4 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
      __builtin_memcpy(&nx, rd.s + pn, 4);
    } else {
      if (pn + 4 <= rd.end) __builtin_memcpy(&nx, rd.s + pn, 4);
    }
  };
  fetch();
  auto refill = [&]() {
    if constexpr (UNST) {
      if (nb <= 32) {
        buf |= (uint64_t)__builtin_bswap32(nx & nm) << (32 - nb);
        nb += 32;
        pn += 4;
        __builtin_amdgcn_sched_barrier(0);      // the old dword is dead before the new one is requested: one register, no copy (and no wait) behind the load
        fetch();
      }
      return;
    }
    while (nb <= 32) {
      const bool have4 = pn + 4 <= rd.end;
      if (have4) {
        const uint32_t x = nx, y = ~x;
        if ((((y - 0x01010101u) & ~y) & 0x80808080u) == 0) {      // no FF among the four bytes
          buf |= (uint64_t)__builtin_bswap32(x) << (32 - nb);
          nb += 32;
          pn += 4;
          fetch();
          continue;
        }
      }
      const uint32_t b = have4 ? (nx & 0xFFu) : rd.ld(pn);
      const uint32_t b1 = have4 ? ((nx >> 8) & 0xFFu) : rd.ld(pn + 1);
      const uint32_t stuffed = (b == 0xFFu && b1 == 0u) ? 1u : 0u;
      sm |= stuffed << ((nb + 7) >> 3);
      buf |= (uint64_t)b << (56 - nb);
      nb += 8;
      pn += 1 + stuffed;
      fetch();
    }
  };
  auto consume = [&](int nbits) {      // drop nbits from the top of the buffer
    if constexpr (UNST) { buf <<= nbits; nb -= nbits; return; }
    const int before = (nb + 7) >> 3;
    buf <<= nbits;
    nb -= nbits;
    const int c = before - ((nb + 7) >> 3);          // bytes that left the buffer
    pb += (uint32_t)c + (uint32_t)__builtin_popcount(sm & ((1u << c) - 1u));
    sm >>= c;
  };
  refill();
  if constexpr (UNST) refill();      // (64 bits buffered, as the stuffed form's loop leaves them)
  consume((int)cb);
  // UNST: rem = bits between the first unread bit and the end of the subsequence (the loop's exit test is one subtraction and a
  // sign test per symbol instead of rebuilding the byte position).  The loop exists twice: when every lane of the wave has its
  // tables in the workgroup's LDS copy - nearly always - the per-lane choice between the LDS and the global lookup (two exec-mask
  // branches per symbol) is not in it.
  int rem = UNST ? (int)(stop * 8u) - (int)(pn * 8u - (uint32_t)nb) : 0;
  const bool wave_lds = __all(in_lds ? 1 : 0) != 0;
  auto symbols = [&](auto lds_only_tag) __attribute__((always_inline)) {
  constexpr bool LDS_ONLY = decltype(lds_only_tag)::value;
  for (;;) {
    if constexpr (UNST) { if (rem <= 0) break; }
    else if (pb >= stop) break;
    if (WRITE && first_block + nblk >= max_blocks) break;     // the rest of the segment is padding
    if (nb < 32) refill();
    // one symbol, DC difference (F.2.2.1) and AC coefficient (F.2.2.2) through the same straight-line code: the lanes of a
    // wave are at different places of their blocks, a branch per symbol kind would run both sides for every symbol
    const bool dc = k == 0;
    const int table = (int)tbl;
    const uint32_t fidx = table * FAST_SIZE + (uint32_t)(buf >> (64 - FAST_BITS));
    // The two lookups in GLOBAL memory (a thread of another table set; a code longer than FAST_BITS) are waited for inside
    // their branches: left to hipcc the wait sits behind the join as vmcnt(0), in every symbol step, and also waits for the
    // dword that the last refill requested for the NEXT refill - the one load that is meant to stay in flight.
    uint32_t e;
    if (LDS_ONLY || in_lds) {
      e = ((const __attribute__((address_space(3))) uint16_t *)sh_fast)[fidx];
    } else {
      e = fast[fidx];
      asm volatile("" : "+v"(e));
    }
    if (e == 0) {
      e = lut_base[(size_t)table * LUT_SIZE + (uint32_t)(buf >> 48)];
      asm volatile("" : "+v"(e));
      if (e == 0) {           // no such code: a wrong-state decoder steps on one bit (as an empty DC / an EOB), the true decode is corrupt
        if (WRITE) atomicExch(err, 1);
        e = lut_entry(1, 0, dc ? 1 : 64);
      }
    }
    // entry = code length | size of the value | advance of the zig-zag index (lut_entry): nothing below depends on the symbol kind
    const int nbits = (int)(e & 31u), s = (int)((e >> 5) & 15u), d = (int)(e >> 9);
    [[maybe_unused]] const int len = nbits - s;
    const int kk = k + d - 1;                                       // zig-zag index the value (if any) belongs to
    if (WRITE && s) {
      if (kk > 63) {
        atomicExch(err, 1);
      } else {
        const int x = (int)((buf << len) >> (64 - s));
        const int val = x < (1 << (s - 1)) ? x - (1 << s) + 1 : x;   // EXTEND (F.2.2.1)
        const uint32_t nat = (uint32_t)kk;      // (the coefficient array is in ZIG-ZAG order since round 5: the IDCT undoes it with compile-time indices;
                                                // the table lookup here was a vector load from constant memory - and its wait - per value)
        if (own || dc) {
          const uint32_t w = nat >> 1, phys = (((w >> 2) ^ sw) << 2) | (w & 3u);
          ((uint16_t *)blk)[phys * 2 + (nat & 1u)] = (uint16_t)(int16_t)val;
        } else {
          coef[(size_t)(first_block + nblk) * 64 + nat] = (int16_t)val;
        }
      }
    }
    // next index: behind the value; ZRL skips 16; EOB ends the block (advance 64)
    const int kn = k + d;
    const bool block_end = kn > 63;
    if (WRITE) {
      own = own || dc;
      if (block_end && own) {
        uint4 *dst = (uint4 *)(coef + (size_t)(first_block + nblk) * 64);
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
          dst[i ^ sw] = *(const uint4 *)(blk + 4 * i);
          *(uint4 *)(blk + 4 * i) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
      if (block_end) own = false;
    }
    k = block_end ? 0 : kn;
    rot = block_end ? ((rot >> 4) | ((rot & 15u) << rot_sh)) : rot;
    tbl = block_end ? (rot & 3u) : (4u | ((rot >> 2) & 3u));      // the NEXT symbol's table, known before its lookup address is formed
    nblk += block_end ? 1u : 0u;
    consume(nbits);           // <= 27 bits, nb >= 32
    if constexpr (UNST) rem -= nbits;
  }
  };
  if (wave_lds) symbols(std::true_type{}); else symbols(std::false_type{});
  if constexpr (UNST) pb = (pn * 8u - (uint32_t)nb) >> 3;
  cp = pb;
  cb = (8u - ((uint32_t)nb & 7u)) & 7u;
  if (WRITE && own) {      // the last block began here and ends in the next subsequence: its values so far, one by one
    for (uint32_t h = 0; h < 64; ++h) {
      const uint32_t w = h >> 1, phys = (((w >> 2) ^ sw) << 2) | (w & 3u);
      const uint16_t v = ((const uint16_t *)blk)[phys * 2 + (h & 1u)];
      if (v) coef[(size_t)(first_block + nblk) * 64 + h] = (int16_t)v;
    }
  }
  slot = (int)(((uint32_t)slot0 + nblk) % (uint32_t)g.bpm);
  SubRec r;
  r.pos = cp * 8 + cb;
  r.state = (uint16_t)(slot * 64 + k);
  r.in_state = 0;
  r.nblk = nblk;
  r.in_pos = 0xFFFFFFFFu;
  return r;
}

// The first-level tables (8 x 1024 entries, 16 KiB) of the table set most of a workgroup's threads need are staged in
// LDS: one dependent load per symbol then costs an LDS round trip instead of an L2 one.  Threads of another set (a
// workgroup that straddles two files with different Huffman tables) read theirs from global memory.
__device__ __forceinline__ void stage_fast(const uint16_t *__restrict__ fast, uint32_t block_set, uint16_t *sh_fast) {
  const uint4 *src = (const uint4 *)(fast + (size_t)block_set * 8 * FAST_SIZE);
  for (int i = threadIdx.x; i < 8 * FAST_SIZE / 8; i += 256) ((uint4 *)sh_fast)[i] = src[i];
  __syncthreads();
}

__device__ __forceinline__ int find_seg(const Seg *__restrict__ segs, int nseg, uint32_t sub) {
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].first_sub <= sub) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------------ un-stuffing (round 5)
// FF 00 -> FF over every segment, in three launches: (1) one thread per 256 input bytes counts the bytes it keeps (a 00 goes iff the
// byte before it is FF - the pairs cannot overlap); (2) exclusive sum of the counts inside each segment (jpeg_block_scan_kernel's
// loop); (3) the same threads write their kept bytes behind each other, from the segment's own first byte on - the un-stuffed
// segment is shorter than the stuffed one, so the copy keeps the batch buffer's layout and a segment's subsequences (counted from
// the stuffed length) cover it with room to spare; the new end goes into a second Seg array, which the decode kernels read.
// 0x80 in every byte of x that is zero (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {
  const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}
// flags (0x80 per byte) of the stuffed zeros among the four bytes of x; ff = 0x80 if the byte before x is an FF (updated)
__device__ __forceinline__ uint32_t stuffed_bytes(uint32_t x, uint32_t &ff) {
  const uint32_t isff = zero_bytes(~x);
  const uint32_t st = zero_bytes(x) & ((isff << 8) | ff);
  ff = (isff >> 24) & 0x80u;
  return st;
}
// sixteen lanes per 256-byte subsequence, 16 bytes each: a wave reads 1 KiB of the stream per instruction.  (One thread per
// subsequence walking its 256 bytes touched 64 lines per load: 0.15 + 0.49 ms per 256 frames for the two kernels, most of what
// the un-stuffed reader saves.)
constexpr int UL = SUBSEQ / 16;      // lanes per subsequence in the un-stuffing kernels (a power of two, 4 .. 64)
static_assert(UL >= 4 && UL <= 64 && (UL & (UL - 1)) == 0, "TN_JPEG_SUBSEQ");
struct UnstLane { uint32_t x[4]; uint32_t st[4]; uint32_t n; bool live; };
__device__ __forceinline__ UnstLane unstuff_lane(const uint8_t *__restrict__ s, const Seg &sg, uint32_t sub, uint32_t part) {
  UnstLane r;
  r.n = 0;
  const uint32_t p0 = sg.byte_start + (sub - sg.first_sub) * SUBSEQ + part * 16u;
  r.live = p0 < sg.byte_end;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r.x[i] = 0; r.st[i] = 0; }
  if (!r.live) return r;
  uint32_t ff = (p0 > sg.byte_start && s[p0 - 1] == 0xFFu) ? 0x80u : 0u;
  if (p0 + 16 <= sg.byte_end) {
    __builtin_memcpy(r.x, s + p0, 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r.st[i] = stuffed_bytes(r.x[i], ff);
      r.n += 4u - (uint32_t)__builtin_popcount(r.st[i]);
    }
  } else {                         // the segment's last bytes: what lies behind the end counts as stuffed (= not kept)
    for (uint32_t q = 0; q < 16; ++q) {
      const uint32_t p = p0 + q;
      const uint32_t b = p < sg.byte_end ? s[p] : 0u;
      const bool drop = p >= sg.byte_end || (b == 0u && ff);
      r.x[q >> 2] |= b << (8 * (q & 3));
      r.st[q >> 2] |= drop ? 0x80u << (8 * (q & 3)) : 0u;
      r.n += drop ? 0u : 1u;
      ff = b == 0xFFu ? 0x80u : 0u;
    }
  }
  return r;
}
__global__ __launch_bounds__(256) void jpeg_unstuff_count_kernel(const uint8_t *__restrict__ scan, const FrameDev *__restrict__ frames,
                                                                 const Seg *__restrict__ segs, int nseg, uint32_t total_sub, uint32_t *__restrict__ kept) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x, sub = gid / UL, part = gid % UL;
  uint32_t n = 0;
  if (sub < total_sub) {
    const Seg sg = segs[find_seg(segs, nseg, sub)];
    n = unstuff_lane(scan + frames[sg.frame].scan_off, sg, sub, part).n;
  }
#pragma unroll
  for (int d = 1; d < UL; d <<= 1) n += __shfl_xor(n, d, UL);
  if (sub < total_sub && part == 0) kept[sub] = n;
}
// exclusive sum of kept[] inside each segment (one workgroup per segment) and the segment with its new end
__global__ __launch_bounds__(256) void jpeg_unstuff_scan_kernel(const Seg *__restrict__ segs, const uint32_t *__restrict__ kept,
                                                                uint32_t *__restrict__ base, Seg *__restrict__ segs_out) {
  __shared__ uint32_t sh[256];
  const Seg sg = segs[blockIdx.x];
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < sg.nsub; c0 += 256) {
    const uint32_t i = c0 + threadIdx.x;
    const uint32_t v = i < sg.nsub ? kept[sg.first_sub + i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const uint32_t a = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < sg.nsub) base[sg.first_sub + i] = carry + sh[threadIdx.x] - v;
    carry += sh[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Seg r = sg;
    r.byte_end = sg.byte_start + carry;
    segs_out[blockIdx.x] = r;
  }
}
__global__ __launch_bounds__(256) void jpeg_unstuff_write_kernel(const uint8_t *__restrict__ scan, const FrameDev *__restrict__ frames,
                                                                 const Seg *__restrict__ segs, int nseg, uint32_t total_sub,
                                                                 const uint32_t *__restrict__ base, uint8_t *__restrict__ out) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x, sub = gid / UL, part = gid % UL;
  const bool in = sub < total_sub;
  Seg sg;
  UnstLane r;
  r.n = 0; r.live = false;
  if (in) {
    sg = segs[find_seg(segs, nseg, sub)];
    r = unstuff_lane(scan + frames[sg.frame].scan_off, sg, sub, part);
  }
  uint32_t pre = r.n;                                    // inclusive sum over the sixteen lanes of the subsequence
#pragma unroll
  for (int d = 1; d < UL; d <<= 1) {
    const uint32_t v = __shfl_up(pre, d, UL);
    if (part >= (uint32_t)d) pre += v;
  }
  if (!in || !r.live || r.n == 0) return;
  uint8_t *o = out + frames[sg.frame].scan_off + sg.byte_start + base[sub] + (pre - r.n);
  if (r.n == 16) {                                       // nothing stuffed among the lane's bytes (dword stores need no alignment here)
    __builtin_memcpy(o, r.x, 16);
    return;
  }
  for (uint32_t q = 0; q < 16; ++q)
    if (!((r.st[q >> 2] >> (8 * (q & 3))) & 0x80u)) *o++ = (uint8_t)(r.x[q >> 2] >> (8 * (q & 3)));
}

// pass 0: from the subsequence's own first byte with a guessed state; pass > 0: from the predecessor's record, repeated
// inside the workgroup (its 256 subsequences are neighbours) until a round changes nothing there, at most INNER times;
// the host launches passes until one stores nothing at all.
constexpr int INNER = 32;
template <bool UNST>
__global__ __launch_bounds__(256) void jpeg_sync_kernel(const uint8_t *__restrict__ scan, const FrameDev *__restrict__ frames,
                                                        const Seg *__restrict__ segs, int nseg, Geom g,
                                                        const uint16_t *__restrict__ luts, const uint16_t *__restrict__ fast, SubRec *rec, uint32_t total_sub, int pass,
                                                        int *__restrict__ changed) {
  const uint32_t sub = blockIdx.x * 256u + threadIdx.x;
  const bool live = sub < total_sub;
  Seg sg;
  uint32_t t = 0;
  if (live) {
    sg = segs[find_seg(segs, nseg, sub)];
    t = sub - sg.first_sub;
  }
  const FrameDev &f = frames[live ? sg.frame : 0];
  __shared__ __attribute__((aligned(16))) uint16_t sh_fast[8 * FAST_SIZE];
  __shared__ uint32_t sh_set;
  if (threadIdx.x == 0) sh_set = f.lut;      // (thread 0 of a launched workgroup is always live)
  __syncthreads();
  stage_fast(fast, sh_set, sh_fast);
  const bool in_lds = f.lut == sh_set;
  fast += (size_t)f.lut * 8 * FAST_SIZE;
  Reader rd{scan + f.scan_off, live ? sg.byte_end : 0u};
  const uint32_t sub_start = live ? sg.byte_start + t * SUBSEQ : 0u, sub_end = sub_start + SUBSEQ;
  if (pass == 0) {
    if (!live) return;
#if TN_JPEG_RUNIN
    // run-in: start TN_JPEG_RUNIN subsequences EARLIER with the guessed state, so that the decoder has usually fallen into step
    // with the true one when it crosses into its own subsequence; the crossing state is recorded as the start of the record (the
    // fix-up pass decodes again only where the predecessor's end differs from it)
    if (t > 0) {
      const bool from_start = t <= (uint32_t)TN_JPEG_RUNIN;        // the segment's first byte: the true state
      uint32_t cp = from_start ? sg.byte_start : sub_start - (uint32_t)TN_JPEG_RUNIN * SUBSEQ;
      if (!UNST && !from_start && rd.ld(cp - 1) == 0xFFu && rd.ld(cp) == 0u) cp += 1;
      const SubRec r0 = decode_sub<false, UNST>(rd, cp, 0, sub_start, 0, 0, g, f, luts, fast, sh_fast, in_lds, nullptr, 0, 0, nullptr);
      SubRec r1 = decode_sub<false, UNST>(rd, r0.pos >> 3, r0.pos & 7, sub_end, r0.state >> 6, r0.state & 63, g, f, luts, fast, sh_fast, in_lds, nullptr, 0, 0, nullptr);
      r1.in_pos = r0.pos;
      r1.in_state = r0.state;
      rec[sub] = r1;
      return;
    }
#endif
    uint32_t cp = sub_start;
    if (!UNST && t > 0 && rd.ld(cp - 1) == 0xFFu && rd.ld(cp) == 0u) cp += 1;     // a stuffed byte is not data
    rec[sub] = decode_sub<false, UNST>(rd, cp, 0, sub_end, 0, 0, g, f, luts, fast, sh_fast, in_lds, nullptr, 0, 0, nullptr);
    return;
  }
  volatile SubRec *vrec = rec;
  uint32_t in_pos = 0xFFFFFFFFu, in_state = 0;
  if (live) { in_pos = vrec[sub].in_pos; in_state = vrec[sub].in_state; }
  for (int it = 0; it < INNER; ++it) {
    int ch = 0;
    if (live && t > 0) {        // (a segment's first subsequence started from the true state in pass 0)
      const uint32_t ppos = vrec[sub - 1].pos, pstate = vrec[sub - 1].state;
      if (ppos != in_pos || pstate != in_state) {     // decode again only from a start this thread has not decoded from yet
        in_pos = ppos;
        in_state = pstate;
        const SubRec r = decode_sub<false, UNST>(rd, ppos >> 3, ppos & 7, sub_end, pstate >> 6, pstate & 63, g, f, luts, fast, sh_fast, in_lds, nullptr, 0, 0, nullptr);
        ch = (vrec[sub].pos != r.pos || vrec[sub].state != r.state || vrec[sub].nblk != r.nblk) ? 1 : 0;
        vrec[sub].pos = r.pos; vrec[sub].state = (uint16_t)r.state; vrec[sub].nblk = r.nblk;
        vrec[sub].in_pos = in_pos; vrec[sub].in_state = (uint16_t)in_state;
        if (it == 0 || ch) *changed = 1;              // any decode in a launch means one more launch has to confirm
      }
    }
    if (!__syncthreads_or(ch)) break;
  }
}

// exclusive prefix sum of the block counts inside each segment: one workgroup per segment
__global__ __launch_bounds__(256) void jpeg_block_scan_kernel(const Seg *__restrict__ segs, const SubRec *__restrict__ rec,
                                                              uint32_t *__restrict__ base) {
  __shared__ uint32_t sh[256];
  const Seg sg = segs[blockIdx.x];
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < sg.nsub; c0 += 256) {
    const uint32_t i = c0 + threadIdx.x;
    const uint32_t v = i < sg.nsub ? rec[sg.first_sub + i].nblk : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const uint32_t a = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < sg.nsub) base[sg.first_sub + i] = carry + sh[threadIdx.x] - v;
    carry += sh[255];
    __syncthreads();
  }
}

// The only blocks of the coefficient array that need a zero background are the ones that STRADDLE a subsequence boundary: a block
// that begins and ends inside one subsequence leaves its thread's LDS buffer as eight whole rows, a straddling one is written value
// by value from two (at most three) threads.  One thread per subsequence: if its predecessor ended inside a block, that block -
// number base[sub] of the segment - is cleared.  (Replaces a memset of the whole array: 708 MB per 256 720p frames, 0.15 ms.)
__global__ __launch_bounds__(256) void jpeg_zero_straddle_kernel(const Seg *__restrict__ segs, int nseg, Geom g, const SubRec *__restrict__ rec,
                                                                 const uint32_t *__restrict__ base, uint32_t total_sub, int16_t *__restrict__ coef) {
  const uint32_t sub = blockIdx.x * 256u + threadIdx.x;
  if (sub >= total_sub) return;
  const Seg sg = segs[find_seg(segs, nseg, sub)];
  if (sub == sg.first_sub) return;                       // a segment starts at a block boundary
  if ((rec[sub - 1].state & 63u) == 0) return;           // so does this subsequence
  const uint32_t b0 = base[sub];
  if (b0 >= sg.nblocks) return;
  uint4 *dst = (uint4 *)(coef + ((size_t)sg.frame * g.blocks_per_frame + sg.block_base + b0) * 64);
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = make_uint4(0u, 0u, 0u, 0u);
}

template <bool UNST>
__global__ __launch_bounds__(256) void jpeg_write_kernel(const uint8_t *__restrict__ scan, const FrameDev *__restrict__ frames,
                                                         const Seg *__restrict__ segs, int nseg, Geom g,
                                                         const uint16_t *__restrict__ luts, const uint16_t *__restrict__ fast,
                                                         const SubRec *__restrict__ rec, const uint32_t *__restrict__ base, uint32_t total_sub,
                                                         int16_t *__restrict__ coef, int *__restrict__ err) {
  const uint32_t sub_raw = blockIdx.x * 256u + threadIdx.x;
  const bool live = sub_raw < total_sub;
  const uint32_t sub = live ? sub_raw : total_sub - 1;
  const Seg sg = segs[find_seg(segs, nseg, sub)];
  const FrameDev &f = frames[sg.frame];
  __shared__ __attribute__((aligned(16))) uint16_t sh_fast[8 * FAST_SIZE];
  __shared__ uint32_t sh_set;
  __shared__ __attribute__((aligned(16))) uint32_t sh_blk[256 * 32];      // one 8 x 8 block of int16 per thread
  if (threadIdx.x == 0) sh_set = f.lut;
  for (int i = 0; i < 32; ++i) sh_blk[threadIdx.x * 32 + i] = 0u;
  __syncthreads();
  stage_fast(fast, sh_set, sh_fast);
  const bool in_lds = f.lut == sh_set;
  fast += (size_t)f.lut * 8 * FAST_SIZE;
  if (!live) return;
  const uint32_t t = sub - sg.first_sub;
  Reader rd{scan + f.scan_off, sg.byte_end};
  const uint32_t sub_start = sg.byte_start + t * SUBSEQ, sub_end = sub_start + SUBSEQ;
  uint32_t cp = sub_start, cb = 0;
  int slot = 0, k = 0;
  if (t > 0) {
    const SubRec pr = rec[sub - 1];
    cp = pr.pos >> 3;
    cb = pr.pos & 7;
    slot = pr.state >> 6;
    k = pr.state & 63;
  }
  const uint32_t b0 = base[sub];
  // the state must agree with the block count (slot = blocks done mod blocks-per-MCU), or the stream is corrupt
  if (b0 % (uint32_t)g.bpm != (uint32_t)slot) {
    atomicExch(err, 1);
    return;
  }
  if (b0 >= sg.nblocks) return;
  int16_t *cf = coef + ((size_t)sg.frame * g.blocks_per_frame + sg.block_base) * 64;
  const SubRec r = decode_sub<true, UNST>(rd, cp, cb, sub_end, slot, k, g, f, luts, fast, sh_fast, in_lds, cf, b0, sg.nblocks, err, sh_blk + threadIdx.x * 32);
  if (t + 1 == sg.nsub && b0 + r.nblk < sg.nblocks) atomicExch(err, 1);    // data ran out before the last block
}

// DC prediction (F.2.2.1): coef[0] of every block of one (file, component) becomes the running sum of the coded
// differences in scan order, restarting at every restart interval.  One workgroup per (file, component), one MCU per
// thread and round.
__global__ __launch_bounds__(256) void jpeg_dc_scan_kernel(Geom g, int16_t *__restrict__ coef) {
  // (round 5: the segmented scan runs inside the waves - six shuffles, no barrier - and crosses them through four LDS entries;
  // the Hillis-Steele form over 256 LDS entries spent sixteen barriers per 256 MCUs: 267 us per 256 720p frames, all latency)
  __shared__ int sh_s[4];
  __shared__ int sh_f[4];
  const int frame = blockIdx.x / g.ncomp, comp = blockIdx.x % g.ncomp;
  int16_t *cf = coef + (size_t)frame * g.blocks_per_frame * 64;
  const int nmcu = g.mcux * g.mcuy;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int carry = 0;
  for (int c0 = 0; c0 < nmcu; c0 += 256) {
    const int m = c0 + threadIdx.x;
    int sum = 0;
    if (m < nmcu)
      for (int sl = 0; sl < g.bpm; ++sl)
        if (g.slot_comp[sl] == comp) sum += cf[((size_t)m * g.bpm + sl) * 64];
    const int flag = (m < nmcu && g.ri > 0 && m % g.ri == 0) ? 1 : 0;
    // inclusive segmented scan over the wave: (s, f) of the lanes before this one; a flag stops the sum
    int s = sum, f = flag;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int as = __shfl_up(s, d, 64), af = __shfl_up(f, d, 64);
      if (lane >= d) {
        if (!f) s += as;
        f |= af;
      }
    }
    if (lane == 63) { sh_s[wv] = s; sh_f[wv] = f; }
    __syncthreads();
    // what enters this wave: the carry of the previous rounds through the waves before it
    int in = carry;
    for (int w = 0; w < wv; ++w) in = sh_f[w] ? sh_s[w] : in + sh_s[w];
    int tot = carry;                                     // ... and through all four: the next round's carry
    for (int w = 0; w < 4; ++w) tot = sh_f[w] ? sh_s[w] : tot + sh_s[w];
    const int incl = f ? s : s + in;                     // inclusive value of this MCU
    if (m < nmcu) {
      int run = flag ? 0 : incl - sum;                   // what precedes this MCU's own blocks
      for (int sl = 0; sl < g.bpm; ++sl)
        if (g.slot_comp[sl] == comp) {
          int16_t *b = cf + ((size_t)m * g.bpm + sl) * 64;
          run += b[0];
          b[0] = (int16_t)run;
        }
    }
    carry = tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ IDCT
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jidctint.c::jpeg_idct_islow, one 1-D pass over 8 values
__device__ __forceinline__ void idct8(const int *in, int *out, int shift) {
  constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  int z2 = in[2], z3 = in[6];
  int z1 = (z2 + z3) * F0_541;
  int tmp2 = z1 + z3 * (-F1_847);
  int tmp3 = z1 + z2 * F0_765;
  int tmp0 = (in[0] + in[4]) * 8192;
  int tmp1 = (in[0] - in[4]) * 8192;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * F1_175;
  tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
  z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
  out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
  out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
  out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}

// one thread per block: dequantise, columns then rows, range limit, store 8 x 8 bytes into the component plane
__global__ __launch_bounds__(64) void jpeg_idct_kernel(Geom g, const FrameDev *__restrict__ frames,
                                                       const int16_t *__restrict__ coef, uint8_t *__restrict__ planes) {
  const int blk = blockIdx.x * 64 + threadIdx.x, frame = blockIdx.y;
  if (blk >= g.blocks_per_frame) return;
  const int m = blk / g.bpm, sl = blk - m * g.bpm, comp = g.slot_comp[sl];
  const int my = m / g.mcux, mx = m - my * g.mcux;
  const int brow = g.ncomp == 1 ? my : my * g.vs[comp] + g.slot_by[sl], bcol = g.ncomp == 1 ? mx : mx * g.hs[comp] + g.slot_bx[sl];
  const int16_t *c = coef + ((size_t)frame * g.blocks_per_frame + blk) * 64;
  const uint16_t *q = frames[frame].q[comp];
  // the block as the write pass left it: 128 bytes in zig-zag order (T.81 figure A.6); kZzInv[natural position] = zig-zag index
  constexpr int kZzInv[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                              10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
  uint4 rows[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) rows[r] = ((const uint4 *)c)[r];
  int16_t cz[64];
  __builtin_memcpy(cz, rows, 128);
  int ws[64];
#pragma unroll
  for (int col = 0; col < 8; ++col) {
    int in[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = (int)cz[kZzInv[r * 8 + col]] * (int)q[r * 8 + col];
    idct8(in, o, 13 - 2);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r * 8 + col] = o[r];
  }
  uint8_t *dst = planes + (size_t)frame * g.planes_per_frame + g.plane_off[comp] + ((size_t)brow * 8) * g.plane_w[comp] + bcol * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int o[8];
    idct8(ws + r * 8, o, 13 + 2 + 3);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = o[i] & 1023;      // jdmaster.c::prepare_range_limit_table behind the IDCT's index mask
      const uint32_t s = v < 128 ? v + 128 : v < 512 ? 255 : v < 896 ? 0 : v - 896;
      if (i < 4) lo |= s << (8 * i); else hi |= s << (8 * (i - 4));
    }
    *(uint2 *)(dst + (size_t)r * g.plane_w[comp]) = make_uint2(lo, hi);
  }
}

// ------------------------------------------------------------------------------------------------ upsample + colour
// jdsample.c fancy upsampling at output position (y, x) of a component subsampled by (hsub, vsub) in {1,2}
__device__ __forceinline__ int upsampled(const uint8_t *__restrict__ p, int pw, int cw, int ch, int hsub, int vsub, int y, int x) {
  if (hsub == 1 && vsub == 1) return p[(size_t)y * pw + x];
  if (cw <= 2) return p[(size_t)(y / vsub) * pw + x / hsub];          // jinit_upsampler: box replication
  const int cx = x >> 1, nx = (x & 1) ? (cx + 1 < cw ? cx + 1 : cw - 1) : (cx > 0 ? cx - 1 : 0);
  if (vsub == 1) {          // h2v1_fancy_upsample
    const uint8_t *row = p + (size_t)y * pw;
    return (3 * row[cx] + row[nx] + ((x & 1) ? 2 : 1)) >> 2;
  }
  const int cy = y >> 1, oy = (y & 1) ? (cy + 1 < ch ? cy + 1 : ch - 1) : (cy > 0 ? cy - 1 : 0);   // h2v2_fancy_upsample
  const uint8_t *r0 = p + (size_t)cy * pw, *r1 = p + (size_t)oy * pw;
  const int cs = 3 * r0[cx] + r1[cx], ns = 3 * r0[nx] + r1[nx];
  return (3 * cs + ns + ((x & 1) ? 7 : 8)) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(Geom g, const uint8_t *__restrict__ planes, uint8_t *__restrict__ rgb) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, frame = blockIdx.z;
  if (x >= g.W) return;
  const uint8_t *pl = planes + (size_t)frame * g.planes_per_frame;
  uint8_t *o = rgb + (((size_t)frame * g.H + y) * g.W + x) * 3;
  const int Y = pl[g.plane_off[0] + (size_t)y * g.plane_w[0] + x];
  if (g.ncomp == 1) {
    o[0] = o[1] = o[2] = (uint8_t)Y;
    return;
  }
  const int cb = upsampled(pl + g.plane_off[1], g.plane_w[1], g.cw[1], g.ch[1], g.hmax / g.hs[1], g.vmax / g.vs[1], y, x) - 128;
  const int cr = upsampled(pl + g.plane_off[2], g.plane_w[2], g.cw[2], g.ch[2], g.hmax / g.hs[2], g.vmax / g.vs[2], y, x) - 128;
  // jdcolor.c::build_ycc_rgb_table: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
  const int r = Y + ((91881 * cr + 32768) >> 16);
  const int b = Y + ((116130 * cb + 32768) >> 16);
  const int gg = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
  o[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
  o[1] = (uint8_t)(gg < 0 ? 0 : gg > 255 ? 255 : gg);
  o[2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
}

// 4:2:0 (h2v2 for both chroma planes, the layout video frames come in): four pixels of a row per thread.  The four
// outputs need the chroma column sums 3 * near row + far row of columns cx0-1 .. cx0+2 only (16 byte loads instead of 32),
// luma comes as one dword and the 12 output bytes leave as three dwords.
__global__ __launch_bounds__(256) void jpeg_color420_kernel(Geom g, const uint8_t *__restrict__ planes, uint8_t *__restrict__ rgb) {
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, frame = blockIdx.z;
  if (x0 >= g.W) return;
  const uint8_t *pl = planes + (size_t)frame * g.planes_per_frame;
  const uint32_t y4 = *(const uint32_t *)(pl + g.plane_off[0] + (size_t)y * g.plane_w[0] + x0);
  const int cw = g.cw[1], ch = g.ch[1];
  const int cy = y >> 1, oy = (y & 1) ? (cy + 1 < ch ? cy + 1 : ch - 1) : (cy > 0 ? cy - 1 : 0);
  const int c0 = x0 >> 1;
  int col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 - 1 + i;
    col[i] = c < 0 ? 0 : c > cw - 1 ? cw - 1 : c;
  }
  int up[2][4];      // upsampled Cb / Cr of the four pixels
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    const uint8_t *p = pl + g.plane_off[1 + comp];
    const uint8_t *r0 = p + (size_t)cy * g.plane_w[1 + comp], *r1 = p + (size_t)oy * g.plane_w[1 + comp];
    int cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = 3 * r0[col[i]] + r1[col[i]];
    // pixel x0+2 / x0+3 sit on column c0+1; past the plane's last column (only for pixels past W) the clamp repeats it
    up[comp][0] = (3 * cs[1] + cs[0] + 8) >> 4;
    up[comp][1] = (3 * cs[1] + cs[2] + 7) >> 4;
    up[comp][2] = (3 * cs[2] + cs[1] + 8) >> 4;
    up[comp][3] = (3 * cs[2] + cs[3] + 7) >> 4;
  }
  uint8_t px[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int Y = (y4 >> (8 * i)) & 255, cb = up[0][i] - 128, cr = up[1][i] - 128;
    const int r = Y + ((91881 * cr + 32768) >> 16);
    const int b = Y + ((116130 * cb + 32768) >> 16);
    const int gg = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    px[3 * i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    px[3 * i + 1] = (uint8_t)(gg < 0 ? 0 : gg > 255 ? 255 : gg);
    px[3 * i + 2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
  }
  uint8_t *o = rgb + (((size_t)frame * g.H + y) * g.W + x0) * 3;
  const int nx = g.W - x0 < 4 ? g.W - x0 : 4;
  if (nx == 4 && (((size_t)o) & 3) == 0) {
    uint32_t w[3];
    __builtin_memcpy(w, px, 12);
    ((uint32_t *)o)[0] = w[0]; ((uint32_t *)o)[1] = w[1]; ((uint32_t *)o)[2] = w[2];
  } else {
    for (int i = 0; i < nx * 3; ++i) o[i] = px[i];
  }
}

// The same for a PAIR of rows and eight pixels per thread (the frames of a video: 4:2:0, W a multiple of 8): the rows 2 cy and
// 2 cy + 1 share the near chroma row cy (far rows cy - 1 / cy + 1), a thread needs the chroma columns c0 - 1 .. c0 + 4 of three
// rows (a dword and two bytes each instead of eight byte loads per four pixels), luma comes as two 8-byte loads and the 2 x 24
// output bytes leave as 8-byte stores that are contiguous across the wave.  Threads are dealt over (row pair, column group)
// linearly, so that no lane idles at widths that are not multiples of 2048.  jpeg_color420_kernel had one row and 1024 pixels
// per workgroup: 0.93 ms per 256 720p frames for 1.06 GB of traffic.
__global__ __launch_bounds__(256) void jpeg_color420x2_kernel(Geom g, const uint8_t *__restrict__ planes, uint8_t *__restrict__ rgb) {
  const int tpp = g.W >> 3, pairs = (g.H + 1) >> 1;            // threads per row pair (W % 8 == 0)
  const int idx = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
  if (idx >= tpp * pairs) return;
  const int cy = idx / tpp, tx = idx - cy * tpp, x0 = tx * 8, c0 = tx * 4;
  const uint8_t *pl = planes + (size_t)frame * g.planes_per_frame;
  const int cw = g.cw[1], ch = g.ch[1];
  const int ym = cy > 0 ? cy - 1 : 0, yp = cy + 1 < ch ? cy + 1 : ch - 1;
  // chroma column sums 3 * near + far of the six columns, for the even row (far = cy - 1) and the odd row (far = cy + 1)
  int cs[2][2][6];          // [component][row parity][column]
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    const uint8_t *p = pl + g.plane_off[1 + comp];
    const int pw = g.plane_w[1 + comp];
    int v[3][6];            // rows cy - 1, cy, cy + 1
    const int rows[3] = {ym, cy, yp};
    if (c0 + 4 <= cw - 1 && c0 >= 1) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uint8_t *q = p + (size_t)rows[r] * pw + c0;
        const uint32_t d = *(const uint32_t *)q;
        v[r][0] = q[-1]; v[r][1] = d & 255; v[r][2] = (d >> 8) & 255; v[r][3] = (d >> 16) & 255; v[r][4] = d >> 24; v[r][5] = q[4];
      }
    } else {                 // the plane's edges: columns clamp to 0 .. cw - 1
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int c = c0 - 1 + i;
          v[r][i] = p[(size_t)rows[r] * pw + (c < 0 ? 0 : c > cw - 1 ? cw - 1 : c)];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      cs[comp][0][i] = 3 * v[1][i] + v[0][i];
      cs[comp][1][i] = 3 * v[1][i] + v[2][i];
    }
  }
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int y = 2 * cy + par;
    if (y >= g.H) break;
    const uint2 y8 = *(const uint2 *)(pl + g.plane_off[0] + (size_t)y * g.plane_w[0] + x0);
    uint8_t px[24];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = 1 + (j >> 1);             // the pixel's chroma column c0 + (j >> 1) in cs[..][..][0..5] (index 0 = c0 - 1)
      const int Y = ((j < 4 ? y8.x : y8.y) >> (8 * (j & 3))) & 255;
      const int cb = ((j & 1) ? (3 * cs[0][par][ci] + cs[0][par][ci + 1] + 7) >> 4 : (3 * cs[0][par][ci] + cs[0][par][ci - 1] + 8) >> 4) - 128;
      const int cr = ((j & 1) ? (3 * cs[1][par][ci] + cs[1][par][ci + 1] + 7) >> 4 : (3 * cs[1][par][ci] + cs[1][par][ci - 1] + 8) >> 4) - 128;
      const int r = Y + ((91881 * cr + 32768) >> 16);
      const int b = Y + ((116130 * cb + 32768) >> 16);
      const int gg = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
      px[3 * j] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
      px[3 * j + 1] = (uint8_t)(gg < 0 ? 0 : gg > 255 ? 255 : gg);
      px[3 * j + 2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
    }
    uint8_t *o = rgb + (((size_t)frame * g.H + y) * g.W + x0) * 3;
    if ((((size_t)o) & 7) == 0) {
      uint2 w[3];
      __builtin_memcpy(w, px, 24);
      ((uint2 *)o)[0] = w[0]; ((uint2 *)o)[1] = w[1]; ((uint2 *)o)[2] = w[2];
    } else {
      for (int i = 0; i < 24; ++i) o[i] = px[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------ host: markers
struct HuffSpec { uint8_t counts[16]; uint8_t syms[256]; int nsym; bool present; };

struct Header {
  int W = 0, H = 0, ncomp = 0, ri = 0;
  int cid[3], hs[3], vs[3], tq[3], td[3], ta[3];
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false};
  HuffSpec dc[4], ac[4];
  size_t scan_begin = 0, scan_end = 0;
};

const uint8_t h_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// T.81 Annex B marker segments up to SOS.  Returns "" or the reason the file cannot be decoded here.
std::string parse_header(const uint8_t *d, size_t n, Header &h) {
  for (int i = 0; i < 4; ++i) h.dc[i].present = h.ac[i].present = false;
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return "not a JPEG file (no SOI marker)";
  size_t pos = 2;
  bool have_sof = false;
  for (;;) {
    while (pos < n && d[pos] != 0xFF) ++pos;
    while (pos < n && d[pos] == 0xFF) ++pos;
    if (pos >= n) return "truncated before the scan (no SOS marker)";
    const int m = d[pos++];
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (m == 0xD9) return "EOI before the scan";
    if (pos + 2 > n) return "truncated marker segment";
    const size_t ln = ((size_t)d[pos] << 8) | d[pos + 1];
    if (ln < 2 || pos + ln > n) return "truncated marker segment";
    const uint8_t *s = d + pos + 2;
    const size_t sl = ln - 2;
    if (m == 0xDB) {
      size_t p = 0;
      while (p < sl) {
        const int pq = s[p] >> 4, tq = s[p] & 15;
        ++p;
        if (tq > 3 || p + (pq ? 128 : 64) > sl) return "bad DQT segment";
        for (int i = 0; i < 64; ++i) {
          const int v = pq ? ((s[p + 2 * i] << 8) | s[p + 2 * i + 1]) : s[p + i];
          h.qt[tq][h_zigzag[i]] = (uint16_t)v;
        }
        h.have_qt[tq] = true;
        p += pq ? 128 : 64;
      }
    } else if (m == 0xC4) {
      size_t p = 0;
      while (p < sl) {
        if (p + 17 > sl) return "bad DHT segment";
        const int tc = s[p] >> 4, th = s[p] & 15;
        if (tc > 1 || th > 3) return "bad DHT segment";
        HuffSpec &hs = tc ? h.ac[th] : h.dc[th];
        int tot = 0;
        for (int i = 0; i < 16; ++i) { hs.counts[i] = s[p + 1 + i]; tot += hs.counts[i]; }
        if (tot > 256 || p + 17 + tot > sl) return "bad DHT segment";
        memcpy(hs.syms, s + p + 17, tot);
        hs.nsym = tot;
        hs.present = true;
        p += 17 + tot;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (sl < 6 || s[0] != 8) return "only 8-bit samples are supported";
      h.H = (s[1] << 8) | s[2];
      h.W = (s[3] << 8) | s[4];
      h.ncomp = s[5];
      if (h.ncomp != 1 && h.ncomp != 3) return "only 1- or 3-component images are supported";
      if (sl < (size_t)(6 + 3 * h.ncomp)) return "bad SOF segment";
      for (int i = 0; i < h.ncomp; ++i) {
        h.cid[i] = s[6 + 3 * i];
        h.hs[i] = s[7 + 3 * i] >> 4;
        h.vs[i] = s[7 + 3 * i] & 15;
        h.tq[i] = s[8 + 3 * i];
        if (h.tq[i] > 3) return "bad SOF segment";
      }
      if (h.W <= 0 || h.H <= 0) return "image has no size";
      have_sof = true;
    } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return "unsupported JPEG process SOF" + std::to_string(m - 0xC0) + " (only baseline sequential Huffman is decoded on the device)";
    } else if (m == 0xDD) {
      if (sl < 2) return "bad DRI segment";
      h.ri = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (!have_sof) return "SOS before SOF";
      if (sl < 1 || s[0] != h.ncomp || sl < (size_t)(1 + 2 * h.ncomp)) return "multi-scan JPEG files are not supported";
      for (int i = 0; i < h.ncomp; ++i) {
        const int cid = s[1 + 2 * i], tt = s[2 + 2 * i];
        int c = -1;
        for (int j = 0; j < h.ncomp; ++j) if (h.cid[j] == cid) c = j;
        if (c != i) return "scan components out of frame order";
        h.td[c] = tt >> 4;
        h.ta[c] = tt & 15;
        if (h.td[c] > 3 || h.ta[c] > 3 || !h.dc[h.td[c]].present || !h.ac[h.ta[c]].present) return "scan refers to a missing Huffman table";
        if (!h.have_qt[h.tq[c]]) return "frame refers to a missing quantisation table";
      }
      h.scan_begin = pos + ln;
      size_t e = n;
      for (size_t q = n; q >= h.scan_begin + 2; --q)
        if (d[q - 2] == 0xFF && d[q - 1] == 0xD9) { e = q - 2; break; }
      h.scan_end = e;
      if (h.scan_end <= h.scan_begin) return "empty scan";
      return "";
    }
    pos += ln;
  }
}

// T.81 Annex C code assignment -> 16-bit prefix table
void build_lut(const HuffSpec &hs, bool is_dc, uint16_t *lut) {
  memset(lut, 0, LUT_SIZE * sizeof(uint16_t));
  if (!hs.present) return;
  unsigned code = 0;
  int k = 0;
  for (int len = 1; len <= 16; ++len) {
    for (int i = 0; i < hs.counts[len - 1]; ++i, ++k, ++code) {
      if (code >= (1u << len)) return;     // over-subscribed table: the remaining codes do not exist
      const unsigned first = code << (16 - len), cnt = 1u << (16 - len);
      const int sym = hs.syms[k], sz = sym & 15, run = sym >> 4;
      const uint16_t e = lut_entry(len, sz, is_dc ? 1 : (sz ? run + 1 : (run == 15 ? 16 : 64)));
      for (unsigned j = 0; j < cnt; ++j) lut[first + j] = e;
    }
    code <<= 1;
  }
}

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return TN_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 64;
    if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) { tn_set_error("tn_jpeg: out of device memory"); return TN_ERR_NOMEM; }
    cap = want;
    return TN_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return TN_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 64;
    if (hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess) { tn_set_error("tn_jpeg: out of pinned host memory"); return TN_ERR_NOMEM; }
    cap = want;
    return TN_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// Persistent host workers for the staging copy (spawning 8 threads per call costs as much as the copy they do)
struct HostPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::function<void(int)> fn;
  int n_items = 0, next = 0, pending = 0;
  bool stop = false;
  explicit HostPool(int nthreads) {
    for (int t = 0; t < nthreads; ++t) th.emplace_back([this] { work(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv_work.notify_all();
    for (auto &t : th) t.join();
  }
  void work() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || next < n_items; });
      if (stop) return;
      const int i = next++;
      lk.unlock();
      fn(i);
      lk.lock();
      if (--pending == 0) cv_done.notify_all();
    }
  }
  void run(int n, std::function<void(int)> f) {       // f(0) .. f(n-1) on the workers; returns when all are done
    if (n <= 0) return;
    std::unique_lock<std::mutex> lk(mu);
    fn = std::move(f); n_items = n; next = 0; pending = n;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
    n_items = 0;
  }
};

struct tn_jpeg {
  tn_ctx *ctx;
  std::unique_ptr<HostPool> pool;
  PinBuf<uint8_t> h_scan;
  PinBuf<FrameDev> h_frames;
  PinBuf<Seg> h_segs;
  PinBuf<int> h_flags;                  // [0] changed, [1] error
  DevBuf<uint8_t> d_scan, d_planes;
  DevBuf<uint8_t> d_scan_u;             // the batch's entropy-coded bytes without their FF 00 stuffing (same layout)
  DevBuf<FrameDev> d_frames;
  DevBuf<Seg> d_segs, d_segs_u;         // segments as uploaded / with the un-stuffed ends
  DevBuf<SubRec> d_rec;
  DevBuf<uint32_t> d_base, d_kept;
  DevBuf<int16_t> d_coef;
  DevBuf<uint16_t> d_luts, d_fast;
  DevBuf<int> d_flags;
  std::map<std::string, int> lut_index;  // Huffman table set (the four DHT payloads of a file) -> index in d_luts
  std::vector<uint16_t> lut_host, fast_host;   // all table sets, host copy (re-uploaded when a new one appears)
  int last_sync_passes = 0;
};

extern "C" int tn_jpeg_info(const uint8_t *data_host, size_t size, int *width, int *height, int *components, int *h_samp, int *v_samp) {
  TN_REQUIRE(data_host, "tn_jpeg_info: null data");
  Header h;
  const std::string why = parse_header(data_host, size, h);
  if (!why.empty()) { tn_set_error("tn_jpeg_info: " + why); return TN_ERR_INVALID; }
  if (width) *width = h.W;
  if (height) *height = h.H;
  if (components) *components = h.ncomp;
  if (h_samp) *h_samp = h.hs[0];
  if (v_samp) *v_samp = h.vs[0];
  return TN_OK;
}

extern "C" int tn_jpeg_create(tn_ctx *ctx, tn_jpeg **out) {
  TN_REQUIRE(ctx && out, "tn_jpeg_create: null argument");
  TN_ON_DEVICE(ctx->device);
  tn_jpeg *j = new tn_jpeg();
  j->ctx = ctx;
  if (j->h_flags.ensure(2) != TN_OK || j->d_flags.ensure(2) != TN_OK) { delete j; return TN_ERR_NOMEM; }
  *out = j;
  return TN_OK;
}

extern "C" int tn_jpeg_destroy(tn_jpeg *j) {
  if (!j) return TN_OK;
  TnDeviceGuard tn_dg_(j->ctx->device);
  (void)hipStreamSynchronize(j->ctx->stream);
  j->h_scan.release(); j->h_frames.release(); j->h_segs.release(); j->h_flags.release();
  j->d_scan.release(); j->d_planes.release(); j->d_frames.release(); j->d_segs.release(); j->d_rec.release();
  j->d_scan_u.release(); j->d_segs_u.release(); j->d_kept.release();
  j->d_base.release(); j->d_coef.release(); j->d_luts.release(); j->d_fast.release(); j->d_flags.release();
  delete j;
  return TN_OK;
}

extern "C" int tn_jpeg_sync_passes(const tn_jpeg *j) { return j ? j->last_sync_passes : 0; }

extern "C" int tn_jpeg_decode(tn_jpeg *j, const uint8_t *const *data_host, const size_t *sizes, int n, uint8_t *rgb, int *width, int *height) {
  TN_REQUIRE(j && data_host && sizes && rgb, "tn_jpeg_decode: null argument");
  TN_REQUIRE(n > 0 && n <= 65535, "tn_jpeg_decode: batch must be in 1..65535");
  TN_ON_DEVICE(j->ctx->device);
  hipStream_t st = j->ctx->stream;
  static const bool timing = getenv("TN_JPEG_TIMING") != nullptr;      // tuning: host wall clock of the call's phases on stderr
  const auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_prev = tnow();
  double t_phase[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int i) { if (timing) { const auto t = tnow(); t_phase[i] += std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; } };

  // ---- headers ----
  std::vector<Header> hd(n);
  for (int i = 0; i < n; ++i) {
    TN_REQUIRE(data_host[i], "tn_jpeg_decode: null file pointer");
    const std::string why = parse_header(data_host[i], sizes[i], hd[i]);
    if (!why.empty()) { tn_set_error("tn_jpeg_decode: file " + std::to_string(i) + ": " + why); return TN_ERR_INVALID; }
  }
  const Header &h0 = hd[0];
  Geom g;
  memset(&g, 0, sizeof(g));
  g.W = h0.W; g.H = h0.H; g.ncomp = h0.ncomp; g.ri = h0.ri;
  g.hmax = g.vmax = 1;
  for (int c = 0; c < g.ncomp; ++c) {
    g.hs[c] = h0.hs[c]; g.vs[c] = h0.vs[c];
    if (g.hs[c] < 1 || g.hs[c] > 4 || g.vs[c] < 1 || g.vs[c] > 4) { tn_set_error("tn_jpeg_decode: bad sampling factors"); return TN_ERR_INVALID; }
    g.hmax = std::max(g.hmax, g.hs[c]); g.vmax = std::max(g.vmax, g.vs[c]);
  }
  if (g.ncomp == 1) {
    g.hs[0] = g.vs[0] = g.hmax = g.vmax = 1;      // a single-component scan is not interleaved (A.2.2)
  } else {
    if (g.hs[0] != g.hmax || g.vs[0] != g.vmax) { tn_set_error("tn_jpeg_decode: luma must carry the maximum sampling factors"); return TN_ERR_INVALID; }
    for (int c = 1; c < 3; ++c) {
      const int hsub = g.hmax / g.hs[c], vsub = g.vmax / g.vs[c];
      const bool ok = g.hmax % g.hs[c] == 0 && g.vmax % g.vs[c] == 0 && ((hsub == 1 && vsub == 1) || (hsub == 2 && vsub == 1) || (hsub == 2 && vsub == 2));
      if (!ok) { tn_set_error("tn_jpeg_decode: unsupported chroma subsampling (4:4:4, 4:2:2 and 4:2:0 are decoded)"); return TN_ERR_INVALID; }
    }
  }
  g.mcux = (g.W + 8 * g.hmax - 1) / (8 * g.hmax);
  g.mcuy = (g.H + 8 * g.vmax - 1) / (8 * g.vmax);
  g.bpm = 0;
  for (int c = 0; c < g.ncomp; ++c)
    for (int by = 0; by < g.vs[c]; ++by)
      for (int bx = 0; bx < g.hs[c]; ++bx) {
        if (g.bpm >= 8) { tn_set_error("tn_jpeg_decode: more than 8 blocks per MCU"); return TN_ERR_INVALID; }      // (decode_sub keeps the slots' table ids in 32 bits; no accepted chroma layout has more)
        g.slot_comp[g.bpm] = c; g.slot_by[g.bpm] = by; g.slot_bx[g.bpm] = bx;
        ++g.bpm;
      }
  g.blocks_per_frame = g.mcux * g.mcuy * g.bpm;
  long off = 0;
  for (int c = 0; c < g.ncomp; ++c) {
    g.plane_w[c] = g.mcux * g.hs[c] * 8;
    g.plane_h[c] = g.mcuy * g.vs[c] * 8;
    g.plane_off[c] = off;
    off += (long)g.plane_w[c] * g.plane_h[c];
    g.cw[c] = (g.W * g.hs[c] + g.hmax - 1) / g.hmax;
    g.ch[c] = (g.H * g.vs[c] + g.vmax - 1) / g.vmax;
  }
  g.planes_per_frame = (off + 15) & ~15L;
  for (int i = 1; i < n; ++i) {
    const Header &h = hd[i];
    bool same = h.W == h0.W && h.H == h0.H && h.ncomp == h0.ncomp && h.ri == h0.ri;
    for (int c = 0; same && c < h.ncomp; ++c) same = (h0.ncomp == 1) || (h.hs[c] == h0.hs[c] && h.vs[c] == h0.vs[c]);
    if (!same) { tn_set_error("tn_jpeg_decode: file " + std::to_string(i) + " differs from file 0 in size, sampling or restart interval (one call decodes the frames of one video)"); return TN_ERR_INVALID; }
  }

  // ---- Huffman table sets, scan bytes, segments ----
  size_t scan_total = 0;
  for (int i = 0; i < n; ++i) scan_total += ((hd[i].scan_end - hd[i].scan_begin + SUBSEQ - 1) / SUBSEQ) * SUBSEQ + 16;
  if (scan_total >= (1ull << 29)) { tn_set_error("tn_jpeg_decode: more than 512 MB of entropy-coded data in one call"); return TN_ERR_INVALID; }
  int rc;
  if ((rc = j->h_scan.ensure(scan_total)) || (rc = j->d_scan.ensure(scan_total)) || (rc = j->h_frames.ensure(n)) || (rc = j->d_frames.ensure(n))) return rc;
  std::vector<Seg> segs;
  bool new_luts = false;
  size_t so = 0;
  uint32_t total_sub = 0;
  const int nmcu = g.mcux * g.mcuy;
  for (int i = 0; i < n; ++i) {
    const Header &h = hd[i];
    FrameDev &f = j->h_frames.p[i];
    memset(&f, 0, sizeof(f));
    std::string key;
    for (int t = 0; t < 4; ++t)
      for (const HuffSpec *hs : {&h.dc[t], &h.ac[t]}) {
        key.push_back(hs->present ? 1 : 0);
        if (hs->present) { key.append((const char *)hs->counts, 16); key.append((const char *)hs->syms, hs->nsym); }
      }
    auto it = j->lut_index.find(key);
    if (it == j->lut_index.end()) {
      const int idx = (int)j->lut_index.size();
      j->lut_host.resize((size_t)(idx + 1) * 8 * LUT_SIZE);
      for (int t = 0; t < 4; ++t) {
        build_lut(h.dc[t], true, j->lut_host.data() + ((size_t)idx * 8 + t) * LUT_SIZE);
        build_lut(h.ac[t], false, j->lut_host.data() + ((size_t)idx * 8 + 4 + t) * LUT_SIZE);
      }
      j->fast_host.resize((size_t)(idx + 1) * 8 * FAST_SIZE);
      for (int t = 0; t < 8; ++t) {       // first level: the full table's entry where the code fits into FAST_BITS bits
        const uint16_t *full = j->lut_host.data() + ((size_t)idx * 8 + t) * LUT_SIZE;
        uint16_t *fs = j->fast_host.data() + ((size_t)idx * 8 + t) * FAST_SIZE;
        for (int q = 0; q < FAST_SIZE; ++q) {
          const uint16_t e = full[(size_t)q << (16 - FAST_BITS)];
          fs[q] = (e & 31) - ((e >> 5) & 15) <= FAST_BITS ? e : 0;      // (code length = bits taken - value bits)
        }
      }
      it = j->lut_index.emplace(key, idx).first;
      new_luts = true;
    }
    f.lut = (uint32_t)it->second;
    for (int c = 0; c < g.ncomp; ++c) memcpy(f.q[c], h.qt[h.tq[c]], 128);
    for (int s = 0; s < g.bpm; ++s) { f.slot_dc[s] = (uint8_t)h.td[g.slot_comp[s]]; f.slot_ac[s] = (uint8_t)h.ta[g.slot_comp[s]]; }
    const size_t len = h.scan_end - h.scan_begin;
    f.scan_off = (uint32_t)so;
    f.scan_len = (uint32_t)len;
    const uint8_t *src = data_host[i] + h.scan_begin;
    const size_t padded = ((len + SUBSEQ - 1) / SUBSEQ) * SUBSEQ + 16;
    so += padded;       // (the bytes are copied below, by several threads)
    // segments: the whole scan, or one per restart interval (E.2.4: RSTm between intervals of ri MCUs)
    if (h.ri == 0) {
      Seg sg{(uint32_t)i, 0u, (uint32_t)len, 0u, (uint32_t)g.blocks_per_frame, total_sub, (uint32_t)((len + SUBSEQ - 1) / SUBSEQ)};
      total_sub += sg.nsub;
      segs.push_back(sg);
    } else {
      size_t start = 0;
      int interval = 0;
      const int nint = (nmcu + h.ri - 1) / h.ri;
      for (size_t p = 0; p + 1 <= len && interval < nint; ++p) {
        const bool at_end = p + 1 >= len;
        const bool marker = !at_end && src[p] == 0xFF && src[p + 1] >= 0xD0 && src[p + 1] <= 0xD7;
        if (!marker && !at_end) continue;
        const size_t e = marker ? p : len;
        const int mc = std::min(h.ri, nmcu - interval * h.ri);
        Seg sg{(uint32_t)i, (uint32_t)start, (uint32_t)e, (uint32_t)(interval * h.ri * g.bpm), (uint32_t)(mc * g.bpm), total_sub,
               (uint32_t)std::max<size_t>(1, (e - start + SUBSEQ - 1) / SUBSEQ)};
        total_sub += sg.nsub;
        segs.push_back(sg);
        ++interval;
        start = p + 2;
        ++p;
      }
      if (interval != nint) { tn_set_error("tn_jpeg_decode: file " + std::to_string(i) + ": restart markers do not match the restart interval"); return TN_ERR_INVALID; }
    }
  }
  lap(0);       // headers, tables, segments
  {   // entropy-coded bytes -> pinned staging buffer -> device, in groups of frames: the H2D copy of a group runs while the
      // pool's threads stage the next one (the staging copy and the H2D used to run one after the other: 2.0 + 1.4 ms per
      // 78 MB batch)
    auto copy_one = [&](int i) {
      const FrameDev &f = j->h_frames.p[i];
      const size_t len = f.scan_len, padded = ((len + SUBSEQ - 1) / SUBSEQ) * SUBSEQ + 16;
      memcpy(j->h_scan.p + f.scan_off, data_host[i] + hd[i].scan_begin, len);
      memset(j->h_scan.p + f.scan_off + len, 0, padded - len);
    };
    const bool big = so > (4u << 20) && n >= 8;
    if (big && !j->pool) {
      const unsigned hw = std::thread::hardware_concurrency();
      const char *pe = getenv("TN_JPEG_THREADS");
      j->pool.reset(new HostPool(pe ? std::max(1, atoi(pe)) : std::max(2, std::min(16, (int)(hw ? hw / 4 : 8)))));
    }
    const int ngroups = big && so > (16u << 20) ? 4 : 1;
    for (int gi = 0; gi < ngroups; ++gi) {
      const int ga = (int)((long)n * gi / ngroups), gb = (int)((long)n * (gi + 1) / ngroups);
      if (big) j->pool->run(gb - ga, [&](int k) { copy_one(ga + k); });
      else for (int i = ga; i < gb; ++i) copy_one(i);
      const size_t off_a = j->h_frames.p[ga].scan_off, off_b = gb < n ? (size_t)j->h_frames.p[gb].scan_off : so;
      TN_HIP_CHECK(hipMemcpyAsync(j->d_scan.p + off_a, j->h_scan.p + off_a, off_b - off_a, hipMemcpyHostToDevice, st));
    }
  }
  const int nseg = (int)segs.size();
  if ((rc = j->h_segs.ensure(nseg)) || (rc = j->d_segs.ensure(nseg)) || (rc = j->d_rec.ensure(total_sub)) || (rc = j->d_base.ensure(total_sub)) ||
      (rc = j->d_coef.ensure((size_t)n * g.blocks_per_frame * 64)) || (rc = j->d_planes.ensure((size_t)n * g.planes_per_frame)))
    return rc;
  memcpy(j->h_segs.p, segs.data(), nseg * sizeof(Seg));
  if (new_luts) {
    TN_HIP_CHECK(hipStreamSynchronize(st));       // nothing may still read the old table buffer
    if ((rc = j->d_luts.ensure(j->lut_host.size()))) return rc;
    if ((rc = j->d_fast.ensure(j->fast_host.size()))) return rc;
    TN_HIP_CHECK(hipMemcpyAsync(j->d_luts.p, j->lut_host.data(), j->lut_host.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    TN_HIP_CHECK(hipMemcpyAsync(j->d_fast.p, j->fast_host.data(), j->fast_host.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    TN_HIP_CHECK(hipStreamSynchronize(st));       // (pageable source)
  }
  TN_HIP_CHECK(hipMemcpyAsync(j->d_frames.p, j->h_frames.p, n * sizeof(FrameDev), hipMemcpyHostToDevice, st));
  TN_HIP_CHECK(hipMemcpyAsync(j->d_segs.p, j->h_segs.p, nseg * sizeof(Seg), hipMemcpyHostToDevice, st));
  TN_HIP_CHECK(hipMemsetAsync(j->d_flags.p, 0, 2 * sizeof(int), st));
  static const bool poison = getenv("TN_JPEG_POISON") != nullptr;      // tests: a coefficient nobody writes or clears shows as a wrong pixel
  if (poison) TN_HIP_CHECK(hipMemsetAsync(j->d_coef.p, 0x55, (size_t)n * g.blocks_per_frame * 64 * sizeof(int16_t), st));

  lap(1);       // staging + issue of the copies
  if (timing) { (void)hipStreamSynchronize(st); lap(2); }      // what of the H2D was still outstanding
  // ---- Huffman decode ----
  const unsigned gsub = (total_sub + 255) / 256;
  static const bool no_unstuff = getenv("TN_JPEG_NO_UNSTUFF") != nullptr;      // A/B runs: the reader that skips FF 00 itself
  const uint8_t *scan_d = j->d_scan.p;
  const Seg *segs_d = j->d_segs.p;
  if (!no_unstuff) {
    if ((rc = j->d_scan_u.ensure(scan_total)) || (rc = j->d_segs_u.ensure(nseg)) || (rc = j->d_kept.ensure(total_sub))) return rc;
    hipLaunchKernelGGL(jpeg_unstuff_count_kernel, dim3((total_sub + 256 / UL - 1) / (256 / UL)), dim3(256), 0, st, j->d_scan.p, j->d_frames.p, j->d_segs.p, nseg, total_sub, j->d_kept.p);
    hipLaunchKernelGGL(jpeg_unstuff_scan_kernel, dim3(nseg), dim3(256), 0, st, j->d_segs.p, j->d_kept.p, j->d_base.p, j->d_segs_u.p);
    hipLaunchKernelGGL(jpeg_unstuff_write_kernel, dim3((total_sub + 256 / UL - 1) / (256 / UL)), dim3(256), 0, st, j->d_scan.p, j->d_frames.p, j->d_segs.p, nseg, total_sub, j->d_base.p, j->d_scan_u.p);
    scan_d = j->d_scan_u.p;
    segs_d = j->d_segs_u.p;
  }
  auto launch_sync = [&](int ps) {
    if (no_unstuff) hipLaunchKernelGGL(jpeg_sync_kernel<false>, dim3(gsub), dim3(256), 0, st, scan_d, j->d_frames.p, segs_d, nseg, g, j->d_luts.p, j->d_fast.p, j->d_rec.p, total_sub, ps, j->d_flags.p);
    else hipLaunchKernelGGL(jpeg_sync_kernel<true>, dim3(gsub), dim3(256), 0, st, scan_d, j->d_frames.p, segs_d, nseg, g, j->d_luts.p, j->d_fast.p, j->d_rec.p, total_sub, ps, j->d_flags.p);
  };
  launch_sync(0);
  int pass = 1;
  for (;; ++pass) {
    if (pass > MAX_SYNC) { tn_set_error("tn_jpeg_decode: the Huffman streams did not synchronise (corrupt data)"); return TN_ERR_INVALID; }
    TN_HIP_CHECK(hipMemsetAsync(j->d_flags.p, 0, sizeof(int), st));
    launch_sync(pass);
    TN_HIP_CHECK(hipMemcpyAsync(j->h_flags.p, j->d_flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
    TN_HIP_CHECK(hipStreamSynchronize(st));
    if (!j->h_flags.p[0]) break;
  }
  j->last_sync_passes = pass;
  lap(3);       // sync passes (each ends in a host round trip)
  hipLaunchKernelGGL(jpeg_block_scan_kernel, dim3(nseg), dim3(256), 0, st, segs_d, j->d_rec.p, j->d_base.p);
  hipLaunchKernelGGL(jpeg_zero_straddle_kernel, dim3(gsub), dim3(256), 0, st, segs_d, nseg, g, j->d_rec.p, j->d_base.p, total_sub, j->d_coef.p);
  if (no_unstuff) hipLaunchKernelGGL(jpeg_write_kernel<false>, dim3(gsub), dim3(256), 0, st, scan_d, j->d_frames.p, segs_d, nseg, g, j->d_luts.p, j->d_fast.p, j->d_rec.p,
                                     j->d_base.p, total_sub, j->d_coef.p, j->d_flags.p + 1);
  else hipLaunchKernelGGL(jpeg_write_kernel<true>, dim3(gsub), dim3(256), 0, st, scan_d, j->d_frames.p, segs_d, nseg, g, j->d_luts.p, j->d_fast.p, j->d_rec.p,
                          j->d_base.p, total_sub, j->d_coef.p, j->d_flags.p + 1);
  hipLaunchKernelGGL(jpeg_dc_scan_kernel, dim3(n * g.ncomp), dim3(256), 0, st, g, j->d_coef.p);
  // ---- IDCT, upsampling, colour ----
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((g.blocks_per_frame + 63) / 64, n), dim3(64), 0, st, g, j->d_frames.p, j->d_coef.p, j->d_planes.p);
  const bool is420 = g.ncomp == 3 && g.hmax == 2 && g.vmax == 2 && g.hs[1] == 1 && g.vs[1] == 1 && g.hs[2] == 1 && g.vs[2] == 1 && g.cw[1] > 2;
  if (is420 && g.W % 8 == 0)
    hipLaunchKernelGGL(jpeg_color420x2_kernel, dim3(((g.W >> 3) * ((g.H + 1) >> 1) + 255) / 256, n), dim3(256), 0, st, g, j->d_planes.p, rgb);
  else if (is420)
    hipLaunchKernelGGL(jpeg_color420_kernel, dim3((g.W + 1023) / 1024, g.H, n), dim3(256), 0, st, g, j->d_planes.p, rgb);
  else
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((g.W + 255) / 256, g.H, n), dim3(256), 0, st, g, j->d_planes.p, rgb);
  TN_HIP_CHECK(hipGetLastError());
  TN_HIP_CHECK(hipMemcpyAsync(j->h_flags.p + 1, j->d_flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
  TN_HIP_CHECK(hipStreamSynchronize(st));
  lap(4);       // write pass, DC scan, IDCT, colour
  if (timing) fprintf(stderr, "tn_jpeg_decode: host %.2f | staging %.2f | H2D tail %.2f | sync passes %.2f | rest %.2f ms\n", t_phase[0], t_phase[1],
                      t_phase[2], t_phase[3], t_phase[4]);
  if (j->h_flags.p[1]) { tn_set_error("tn_jpeg_decode: corrupt entropy-coded data (bad Huffman code, coefficient index or block count)"); return TN_ERR_INVALID; }
  if (width) *width = g.W;
  if (height) *height = g.H;
  return TN_OK;
}
