// decode.hip -- decompress path on the GPU.
//
//   k_dec_frame    walk the stream framing (signal_packer_base.cpp:98-119,
//                  hzr_decode.c:626-674): plane chunk lengths, hzr block headers
//                  -> per-block input offsets, consumed length, means header
//   k_dec_block    one 1024-thread workgroup per hzr block (hzr_decode.c:335-567): copy / fill /
//                  Huffman+RLE with a 10-bit LUT (the reference uses 8 bits), 32-entry second-level
//                  tables for the prefixes of longer codes and a node walk for what is deeper still;
//                  the code bits are decoded in up to 1024 self-synchronising chunks.
//                  CRCs are checked only on request (rspt_hip_set_verify), as hzr_verify does;
//                  the reference's decoder skips them too (hzr_decode.c:343).
//   k_inv_*        planes -> int32 with sign extension from nb bytes
//                  (signal_packer_base.cpp:121-138), then the inverse xdelta:
//                  inclusive XOR scan, +128, inclusive sum (utils.cpp:204-236)
//                  as a three-pass tiled scan over the flat array
//   k_inv_rows / k_inv_scan_rows / k_inv_native   the same for int32 blocks with ns % 256 == 0: two passes over the planes, the
//                  second one writes the interleaved samples itself
//   k_planar_native   [nch][ns] int32 -> interleaved native (utils.cpp:51-121); k_planar_native_i32x4: the int32 fast path
//                  (every kernel that writes samples reverses their bytes for a big-endian handle: utils.cpp:57-64,77-85,97-104)
#include "common.hpp"

namespace rspt {

constexpr uint64_t kBadBit = 1ull << 63;

__device__ __forceinline__ uint32_t ld_le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld_le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// Where stream b starts and how many bytes it may span: fixed stride, or the (offset, length) index of a container
// (include/rspt_hip.h: 'RSPTPACK'; `src` then points at the container's payload).
constexpr uint64_t kIdxLenMask = (1ull << 56) - 1ull;  // index length word: length | nb << 56 | invalid << 63
__device__ __forceinline__ const uint8_t* stream_base(const uint8_t* src, uint64_t src_stride, const uint64_t* __restrict__ pidx, uint32_t b,
                                                     uint64_t& limit) {
    if (pidx) {
        limit = pidx[2 * b + 1] & kIdxLenMask;
        return src + pidx[2 * b];
    }
    limit = src_stride;
    return src + (size_t)b * src_stride;
}

// one thread per (block, plane).  Container form: the header and the index entry are validated against the container's
// byte length before anything is read through them (a truncated or corrupt gather flags the stream instead of reading
// out of bounds), and the stream's own nb comes from its index entry -- streams of one container may differ (escalation
// inside a batch, shards of independently escalating ranks).  dec_nb[b] = planes of stream b, for the kernels that follow.
__global__ void k_dec_frame(const uint8_t* __restrict__ src, uint64_t src_stride_in, uint32_t nblocks, Geom g, const uint32_t* __restrict__ nb_state,
                            uint64_t* __restrict__ blk_off, uint64_t* __restrict__ consumed, uint8_t* __restrict__ means,
                            const uint64_t* __restrict__ pidx, uint32_t* __restrict__ dec_counter, uint64_t packed_len,
                            uint32_t* __restrict__ dec_nb) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) *dec_counter = 0;  // k_dec_block's work queue
    const uint32_t b = t / kMaxPlanes, k = t % kMaxPlanes;
    if (b >= nblocks) return;
    uint32_t nb = *nb_state;
    bool bad = false;
    if (pidx) {
        const uint64_t* head = pidx - 4;  // magic, nblocks, payload bytes, nb
        const uint64_t payload = head[2];
        const uint64_t off = pidx[2 * b], lw = pidx[2 * b + 1], len = lw & kIdxLenMask;
        // (no sums of untrusted words: the host has checked packed_len >= 32 + 16 * nblocks, so the subtraction cannot wrap,
        //  while `32 + 16 * nblocks + payload` would for a payload word near 2^64)
        if (head[0] != 0x4B43415054505352ull || head[1] != nblocks || payload > packed_len - (32ull + 16ull * nblocks) || (lw >> 63) ||
            off > payload || len > payload - off)
            bad = true;
        const uint32_t nbi = (uint32_t)(lw >> 56) & 0xFu;
        if (nbi >= 1 && nbi <= 4) nb = nbi;  // (0: a container written without per-stream nb -> the handle's state)
    }
    if (k == 0) dec_nb[b] = bad ? 0u : nb;
    if (bad) {
        if (k == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
        for (uint32_t j = 0; j < g.nblk; ++j) blk_off[hb_index(g, b, k, 0) + j] = ~0ull;
        return;
    }
    if (k >= nb) return;
    uint64_t src_stride;  // bytes stream b may span
    const uint8_t* s = stream_base(src, src_stride_in, pidx, b, src_stride);
    uint64_t pos = 1ull + g.hdr_len;
    for (uint32_t kk = 0; kk < k; ++kk) {
        if (pos + 4 > src_stride) {
            bad = true;
            break;
        }
        pos += 4ull + ld_le32(s + pos);
    }
    uint64_t plen = 0;
    if (!bad && pos + 8 <= src_stride) {
        plen = ld_le32(s + pos);
        if (ld_le32(s + pos + 4) != g.N || pos + 4 + plen > src_stride) bad = true;
    } else {
        bad = true;
    }
    const uint32_t hb0 = hb_index(g, b, k, 0);
    if (!bad) {
        uint64_t q = pos + 8;
        const uint64_t pend = pos + 4 + plen;
        for (uint32_t j = 0; j < g.nblk; ++j) {
            if (q + 7 > pend) {
                bad = true;
                break;
            }
            blk_off[hb0 + j] = q;
            const uint32_t L = ld_le16(s + q) + 1u;
            const uint32_t mode = s[q + 6];
            if (mode > 2 || q + 7 + L > pend) {
                bad = true;
                break;
            }
            q += 7ull + L;
        }
        if (!bad && q != pend) bad = true;
    }
    if (bad) {
        for (uint32_t j = 0; j < g.nblk; ++j) blk_off[hb0 + j] = ~0ull;
        atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
    }
    if (k == nb - 1 && !bad) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)(pos + 4 + plen));
    if (k == 0 && 1ull + g.hdr_len <= src_stride)  // (a stream shorter than its own header is flagged above and not read here)
        for (uint32_t i = 0; i < g.hdr_len; ++i) means[(size_t)b * g.hdr_len + i] = s[1 + i];  // base.cpp:111-115
}

// ---------------------------------------------------------------------------
// k_dec_block: one 1024-thread workgroup per hzr block.
//
// A Huffman stream has no index, but a decoder that starts at a wrong bit falls into step with the right one after a
// few symbols.  So the code bits are cut into up to 1024 chunks and every thread decodes one:
//   pass 0     from the chunk's nominal first bit (only chunk 0's is certain) to the first code boundary past its end
//   pass 1..   a chunk whose predecessor ended somewhere else than where it started decodes again from there; the first
//              wrong chunk always becomes right, so the loop ends (typically after 2-3 rounds, at worst one per chunk)
//   final      prefix sum of the bytes each chunk produces, then every chunk decodes once more and writes its literals
//              (the output is zeroed first: zero runs are skips)
// Speculative rounds meet garbage by design: they never report errors, they only stop at the end of the payload.
// ---------------------------------------------------------------------------
constexpr uint32_t kLutBits = 10;
constexpr uint32_t kLutSlow = 0xFFFFFFFFu;
constexpr uint32_t kDecThreads = 1024;
constexpr uint32_t kDecMinChunkBits = 256;
constexpr uint32_t kSerialTreePayload = 4096;  // bytes: below it the tree is recovered by one wave (dec_block)

constexpr uint32_t kNodeSlots = 528;
// Codes longer than the table index.  On the dense plane 0.7 % of the tokens are (11 .. 16 bits), so one lane in 64 meets one in
// every third round of the token loops -- and the whole wave then followed it down the tree, an LDS round trip per level.  The
// prefixes that lead to such codes are few (a handful of 1024): each of the first 32 gets a 32-entry table over the next 5 bits,
// which ends codes up to 15 bits with ONE more read; what is deeper still (or past the 32nd prefix) walks on from there.
constexpr uint32_t kSubBits = 5, kSubSlots = 32;
struct DecLds {
    uint32_t stage[kHzrBlock / 4 + 16];  // payload image; payload byte i sits at byte (skew + i)
    uint32_t lut[1u << kLutBits];        // code of <= 10 bits: tok_entry(sym, len);  longer: kLutLong | (kLutSub | slot, or the node reached after 10 bits)
    uint32_t cend[kDecThreads];          // first code boundary past a chunk's end  (lut + cend: 8 KiB of scratch for the tree parse)
    uint32_t node[kNodeSlots];           // pre-order ids (<= 521 used; walks clamp the id).  leaf: kNodeLeaf | sym;  branch: id of child_b (child_a = id + 1)
    uint32_t lut2[kSubSlots << kSubBits];  // second level: slot s covers the 5 bits behind a 10-bit prefix that is no code yet
    uint32_t slot_node[kSubSlots];         // the node such a prefix leads to
    uint32_t nslot;                        // prefixes that asked for a slot (the first kSubSlots got one)
    uint32_t flag[4];                      // workgroup-wide "did any thread ..." answers (wg_any), zeroed with the staging
    uint32_t wsum[kDecThreads / 64];
    uint32_t nleaf, nnode, endpos;
    uint32_t err;
    uint32_t changed;
    uint32_t code0;  // first bit of the codes
};
constexpr uint32_t kLutLong = 0x80000000u;
constexpr uint32_t kLutSub = 0x40000000u;  // with kLutLong: the low bits are a slot of lut2, not a node
static_assert(sizeof(DecLds) + 64 <= 80 * 1024, "two workgroups per CU");
constexpr uint32_t kNodeLeaf = 0x80000000u;

// 32 stream bits starting at absolute bit position `bp` of the LDS image
__device__ __forceinline__ uint32_t peek32(const uint32_t* st, uint32_t bp) {
    const uint32_t w = bp >> 5, sh = bp & 31u;
    const unsigned long long v = (unsigned long long)st[w] | ((unsigned long long)st[w + 1] << 32);
    return (uint32_t)(v >> sh);
}

// One assembled output dword to memory, byte by byte: for the two dwords at the ends of a chunk's output range, which it may
// share with its neighbours.  A byte that is zero in `acc` is a zero-run byte (already zero in the pre-zeroed output) or a
// neighbour's (which may be writing it right now): only the non-zero bytes are stored.
__device__ __forceinline__ void flush_edge_dword(uint8_t* out, uint32_t dw, uint32_t acc) {
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t v = (acc >> (8 * q)) & 0xFFu;
        if (v) out[dw * 4 + q] = (uint8_t)v;
    }
}

// What the token loop needs to know about a symbol: literal byte (or 256 for any run) | extra bits << 15 | output bytes
// before the extra value << 19 (1 for a literal -- symbol 0 included: one zero byte --, else the run's base length: hzr_internal.h:117-121,
// 2 / 3.. / 7.. / 23.. / 279.. zeros with 0 / 2 / 4 / 8 / 14 extra bits).  A table entry adds (code length + extra bits) << 9: tok_entry.
__device__ __forceinline__ uint32_t tok_meta(uint32_t sym) {
    const bool lit = sym < 256u;
    const uint32_t ri = sym - 256u;  // 0..4 for a run (symbols > 260 never leave the tree parse)
    const uint32_t eb = lit ? 0u : (0xE8420u >> ((ri & 7u) * 4u)) & 15u;
    const uint32_t zb = lit ? 1u : ri == 4u ? 279u : (0x17070302u >> ((ri & 3u) * 8u)) & 255u;
    return (lit ? sym : 256u) | (eb << 15) | (zb << 19);
}
// a table entry: tok_meta | (code length + extra bits = what the token moves the stream position by) << 9
__device__ __forceinline__ uint32_t tok_entry(uint32_t sym, uint32_t len) {
    const uint32_t m = tok_meta(sym);
    return m | ((len + ((m >> 15) & 15u)) << 9);
}
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

enum DecMode { kDecScan = 0, kDecCount = 1, kDecWrite = 2 };  // code boundaries only / + bytes produced / + the bytes themselves

template <int MODE>
__device__ __forceinline__ uint32_t dec_chunk(const DecLds& d, uint32_t bp, uint32_t limit, uint32_t bit_end, uint32_t& produced, uint8_t* out,
                                              uint32_t o0, uint32_t out_size, uint32_t& err, uint32_t max_out = 0xFFFFFFFFu) {
    constexpr bool WRITE = MODE == kDecWrite, COUNT = MODE != kDecScan;
    uint32_t o = o0;
    // WRITE: the aligned output dword being assembled.  The dword that holds o0 may be shared with the chunk in front (and
    // the last one with the chunk behind): those two leave byte by byte, every dword in between belongs to this chunk alone
    // and leaves as one store -- its zero bytes are zero-run bytes, zero in the pre-zeroed output as well.
    const uint32_t first_dw = o0 >> 2;
    uint32_t cur_dw = first_dw, acc = 0;
    uint32_t lim = limit;  // 0 once the lane is done for good (a bad code)
    const uint32_t o_end = WRITE ? o0 + max_out : 0u;  // WRITE: the lane's byte budget ends here (the output position only grows)
    for (;;) {
        // (limit <= bit_end: a lane that ran over the payload is past its limit too -- tested once, behind the loop)
        const bool active = bp < lim && (!WRITE || o < o_end);
        if (!any_lane(active)) break;
        const uint32_t wi = bp >> 5, sh = bp & 31u;
        const uint32_t w0 = d.stage[wi], w1 = d.stage[wi + 1];  // (a done lane reads inside the slack words)
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
        uint32_t e = d.lut[lo & ((1u << kLutBits) - 1u)];
        e = active ? e : 0u;  // an entry of zero moves nothing: the lanes that are done idle through the rest
        if (any_lane((int32_t)e < 0)) {
            // kLutSlow: no such code (a speculative round in the middle of raw bits, or a corrupt stream).  Else a code longer
            // than the table index: walk on from the node the first 10 bits lead to (one LDS read per level), all lanes in
            // step; the others re-read a clamped slot and keep what they have
            bool bad = e == kLutSlow;
            // second level: the next 5 bits pick the entry of the prefix's slot -- a code of 11 .. 15 bits, or the node a longer one goes on from
            const bool sub = !bad && (e & (kLutLong | kLutSub)) == (kLutLong | kLutSub);
            const uint32_t e2 = d.lut2[sub ? ((e & (kSubSlots - 1u)) << kSubBits) | ((lo >> kLutBits) & ((1u << kSubBits) - 1u)) : 0u];
            e = sub ? e2 : e;
            const bool islong = (int32_t)e < 0 && !bad;
            if (any_lane(islong)) {
                uint32_t nd = min(e & 1023u, kNodeSlots - 1u);
                uint32_t wv = d.node[nd], len2 = sub ? kLutBits + kSubBits : kLutBits;
                for (;;) {
                    const bool step = islong && !(wv & kNodeLeaf) && len2 < 32u;
                    if (!any_lane(step)) break;
                    const uint32_t nn = min(((lo >> (len2 & 31u)) & 1u) ? wv : nd + 1u, kNodeSlots - 1u);
                    nd = step ? nn : nd;
                    wv = d.node[nd];
                    len2 += step ? 1u : 0u;
                }
                bad = bad || (islong && !(wv & kNodeLeaf));
                e = islong ? tok_entry(wv & 511u, len2) : e;
            }
            e = bad ? 0u : e;
            lim = bad ? 0u : lim;
            err |= bad ? 1u : 0u;
        }
        const uint32_t adv = (e >> 9) & 63u;  // code + extra bits
        if (COUNT) {
            const uint32_t zb = (e >> 19) & 511u, eb = (e >> 15) & 15u;
            // Extra bits: none in most rounds of a dense block (literals and short runs), so the whole wave tests for them first.
            // They sit in `lo` unless code + extra bits reach past 32 (a deep code in front of a long run's count): then --
            // rarely -- a third image word and a 64-bit shift.
            uint32_t extra = 0;
            if (any_lane(eb != 0u)) {
                const uint32_t len = adv - eb;
                extra = (lo >> (len & 31u)) & ((1u << eb) - 1u);
                if (any_lane(adv > 32u)) {
                    const uint32_t w2 = d.stage[wi + 2];
                    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
                    const unsigned long long win = ((unsigned long long)hi << 32) | lo;
                    extra = (uint32_t)(win >> len) & ((1u << eb) - 1u);
                }
            }
            if (WRITE) {
                // Literals gather in the aligned dword they fall into; the dword leaves once the output position has moved to
                // another one.  Every token takes part -- a run, or the zero entry of an idle lane, just contributes a zero
                // byte -- so there is nothing to select.  (An active lane has o - o0 < max_out = out_size - o0: a literal
                // lands inside the block.)
                const uint32_t dw = o >> 2;
                const bool fl = dw != cur_dw;
                if (any_lane(fl)) {
                    if (fl) {
                        if (cur_dw != first_dw)
                            reinterpret_cast<uint32_t*>(out)[cur_dw] = acc;
                        else
                            flush_edge_dword(out, cur_dw, acc);
                    }
                }
                acc = fl ? 0u : acc;
                cur_dw = dw;
                acc |= (e & 0xFFu) << ((o & 3u) * 8u);
            }
            o += zb + extra;
        }
        bp += adv;
    }
    if (bp > bit_end) err = 1;  // ran over the payload
    if (WRITE) flush_edge_dword(out, cur_dw, acc);
    produced = o - o0;
    return bp;
}

// table entry of a 10-bit prefix that ends at branch `nd`: a slot of the second level while there are any
__device__ __forceinline__ uint32_t long_prefix_entry(DecLds& d, uint32_t nd) {
    const uint32_t sl = atomicAdd(&d.nslot, 1u);
    if (sl >= kSubSlots) return kLutLong | nd;
    d.slot_node[sl] = nd;
    return kLutLong | kLutSub | sl;
}

// Did any thread of the workgroup see `p`?  One barrier; slot `i` is zero when the block starts and used once per block.
// (__syncthreads_or costs three barriers and a cross-lane reduction -- and its one-wave shortcut, never taken here, is a path on
//  which tools/check_barrier_waits.py cannot see the LDS stores in front of it published.)
__device__ __forceinline__ bool wg_any(DecLds& d, uint32_t i, bool p) {
    if (p) d.flag[i] = 1u;
    __syncthreads();
    return d.flag[i] != 0u;
}

__shared__ DecLds g_dec;

// one hzr block (plane k, block j of stream b) by one 1024-thread workgroup
__device__ __forceinline__ void dec_block(uint32_t k, uint32_t j, uint32_t b, const uint8_t* __restrict__ src, uint64_t src_stride, const Geom& g,
                                          const uint32_t* __restrict__ dec_nb, const uint64_t* __restrict__ blk_off,
                                          uint8_t* __restrict__ planes, uint64_t* __restrict__ consumed,
                                          unsigned long long* __restrict__ stamps, const CrcConsts* __restrict__ vcc,
                                          const uint64_t* __restrict__ pidx) {
    DecLds& d = g_dec;
    if (k >= dec_nb[b]) return;
    // (opaque per block: what derives from the thread index -- a few dozen LDS addresses -- is cheap to recompute; hoisted out of
    //  the kernel's persistent loop it sits in scratch memory and comes back by loads inside the barrier-bound tree rounds)
    uint32_t tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const uint32_t tid = tid_, l = tid & 63u, w = tid >> 6;
    const uint32_t hb = hb_index(g, b, k, j);
#define DEC_STAMP(i) do { if (stamps && tid == 0 && hb < 512u) stamps[hb * 8u + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    DEC_STAMP(0);
    // diagnostic: wall-clock (100 MHz) start/end of every workgroup, for a concurrency census (tools/census_decode.py)
    if (stamps && tid == 0 && hb < 16384u) stamps[65536u + 2u * hb] = __builtin_amdgcn_s_memrealtime();
    const uint64_t off = blk_off[hb];
    if (off == ~0ull) return;
    uint64_t lim_unused;
    const uint8_t* s = stream_base(src, src_stride, pidx, b, lim_unused) + off;
    const uint32_t L = ld_le16(s) + 1u;
    const uint32_t mode = s[6];
    const uint32_t out_size = min(kHzrBlock, g.N - j * kHzrBlock);
    uint8_t* out = planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + (size_t)j * kHzrBlock;  // 16-byte aligned

    if (mode == kModeFill) {  // hzr_decode.c:362-370
        if (vcc) {
            // hzr_verify checks the CRC of EVERY block (hzr_decode.c:569-624), a Fill block's one payload byte included: a flip in
            // the fill value or in its CRC, or a mode byte turned into 2, is an error with verification on (found by the damaged-
            // stream leg of tests/soak.py: such streams decoded to other bytes without a word).  Bytewise by one lane: L is 1 for
            // every Fill block an encoder writes; anything longer is not worth more than being correct.
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < L; ++i) c = vcc->table[0][(c ^ s[7 + i]) & 0xFFu] ^ (c >> 8);
            if (~c != ld_le32(s + 2)) {  // (block-uniform: every thread computes the same)
                if (tid == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
                return;
            }
        }
        const uint32_t v = s[7] * 0x01010101u;
        for (uint32_t i = tid; i < (out_size + 15) / 16; i += kDecThreads) reinterpret_cast<uint4*>(out)[i] = make_uint4(v, v, v, v);  // rows are padded
        return;
    }
    // stage the payload with 16-byte aligned global loads
    const uint8_t* pay = s + 7;
    const uint32_t skew = (uint32_t)(reinterpret_cast<uintptr_t>(pay) & 15u);
    const uint8_t* abase = pay - skew;
    for (uint32_t o = tid * 16; o < skew + L; o += kDecThreads * 16)
        *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(d.stage) + o) = *reinterpret_cast<const uint4*>(abase + o);
    if (tid < 4) d.stage[((skew + L + 3) >> 2) + tid] = 0;  // the bit window reads up to two words past the last payload word
    if (tid == 0) d.nslot = 0;
    if (tid < 4) d.flag[tid] = 0;
    __syncthreads();

    if (vcc) {
        // hzr_verify (hzr_decode.c:569-624): CRC-32C of the payload against the header.  Word-strided lanes over the
        // staged payload as in the encoder (hzr_kernels.hip: encode_block), without the X prefix: crc = ~(raw(payload) ^
        // 0xFFFFFFFF * x^(8 L)); the x^(8*4096) step table comes straight from global memory (this path is optional).
        const uint32_t nvw = (L + 3u) >> 2;
        const uint32_t Kst = (nvw + kDecThreads - 1) / kDecThreads;
        const uint8_t* pb = reinterpret_cast<const uint8_t*>(d.stage) + skew;  // payload byte 0
        auto vword = [&](uint32_t r) -> uint32_t {                             // payload bytes [L - 4(r+1), L - 4r), zero in front
            const int32_t lo = (int32_t)L - 4 * (int32_t)(r + 1);
            uint32_t v = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) v |= (lo + q >= 0 ? (uint32_t)pb[lo + q] : 0u) << (8 * q);
            return v;
        };
        uint32_t c = 0;
        if (tid < nvw) {
            for (uint32_t kk = Kst - 1; kk >= 1; --kk) {
                const uint32_t r = tid + kDecThreads * kk;
                if (r < nvw) c ^= vword(r);
                c = gf_shift(vcc, 78, c);  // * x^(8*4096)
            }
            c ^= vword(tid);
            c = gf_shift4(vcc, l, c);  // to the end of the wave's 64 words
        }
        c = wave_xor_u32(c);
        if (l == 0) d.wsum[w] = gf_shift(vcc, 63u - 4u * w, c);  // * x^(8*256*w): to the end of the payload
        __syncthreads();
        uint32_t bad_crc = 0;
        if (tid == 0) {
            uint32_t raw = 0;
            for (uint32_t i = 0; i < kDecThreads / 64; ++i) raw ^= d.wsum[i];
            // the initial state 0xFFFFFFFF travels through L bytes: 4096 a + 64 b + 4 c4 + dbytes
            uint32_t init = 0xFFFFFFFFu;
            const uint32_t a = L >> 12, bq = (L >> 6) & 63u, c4 = (L >> 2) & 15u, dbytes = L & 3u;
            if (a) init = gf_shift(vcc, a == 16 ? 80u : 64u + (15u - a), init);  // x^(8*4096*a); a = 16 only for L = 65536
            if (bq) init = gf_shift(vcc, 63u - bq, init);         // x^(8*64*bq)
            if (c4) init = gf_shift4(vcc, c4 - 1u, init);         // x^(8*4*c4)
            for (uint32_t q = 0; q < dbytes; ++q) init = vcc->table[0][init & 0xFFu] ^ (init >> 8);
            bad_crc = (~(raw ^ init)) != ld_le32(s + 2) ? 1u : 0u;
        }
        if (wg_any(d, 0, bad_crc != 0)) {
            if (tid == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
            return;
        }
    }

    if (mode == kModeCopy) {  // hzr_decode.c:351-359
        if (L != out_size) {
            if (tid == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
            return;
        }
        for (uint32_t i = tid; i < (out_size + 15) / 16; i += kDecThreads) {
            uint32_t wq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t a = skew + i * 16 + q * 4;
                wq[q] = __builtin_amdgcn_alignbyte(d.stage[(a >> 2) + 1], d.stage[a >> 2], a & 3u);
            }
            reinterpret_cast<uint4*>(out)[i] = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        }
        return;
    }

    // ---- Huffman + RLE -------------------------------------------------------
    for (uint32_t i = tid; i < (out_size + 15) / 16; i += kDecThreads) reinterpret_cast<uint4*>(out)[i] = make_uint4(0, 0, 0, 0);  // zero runs = untouched bytes
    const uint32_t bit0 = skew * 8, bit_end = (skew + L) * 8;
    DEC_STAMP(1);
    // Two recoveries of the same tree.  A light block (short payload: a handful of symbols) is parsed by ONE wave on the scalar
    // unit in a few thousand cycles; the workgroup-wide version below it takes ~48 k cycles whatever the tree -- three quarters
    // of a light block's decode time -- but wins on the 200-node trees of dense blocks, where the serial walk (~600 cycles
    // per node: one wave issues an instruction every few cycles) would take 130 k.  The payload length decides; both are exact.
    if (L < kSerialTreePayload) {
    // RecoverTree (hzr_decode.c:263-333).  The description is pre-order: '1' + 9-bit symbol = leaf, '0' = branch followed by
    // child_a then child_b; nodes are numbered in pre-order, so child_a = id + 1 and only child_b has to be stored.
    //   parse   ONE wave walks the description on the scalar unit: its dwords sit in the lanes of two vector registers (fetched
    //           with v_readlane into a 64-bit scalar window), the branches on the path from the root in the lanes of a third
    //           (lane = depth), "already in child_b" as a bit per depth.  A leaf closes every branch whose child_b it ends;
    //           the node behind it is the child_b of the deepest branch still in its child_a.  <= 521 short iterations, no
    //           barrier -- the workgroup-wide pointer-doubling version this replaces took 28 barriers and 48 k cycles per
    //           block, three quarters of a light block's decode time.
    //   table   every thread walks its own 10-bit index down from the root (<= 10 dependent LDS reads, all threads at once):
    //           a leaf on the way -> symbol | length, else the node where longer codes continue.
    constexpr uint32_t kMaxDesc = 11u * kNumSym + 16u;  // description bits (11 S - 1) and a little slack
    const uint32_t P = min(kMaxDesc, bit_end - bit0);
    if (w == 0) {
        const uint32_t w0 = bit0 >> 5;  // first dword of the description in the image
        const uint32_t dv0 = d.stage[w0 + l], dv1 = d.stage[w0 + 64u + l];  // (the image is 64 KiB + slack: always in bounds)
        unsigned long long win = (unsigned long long)read_lane(dv0, 0) >> (bit0 & 31u);
        uint32_t navail = 32u - (bit0 & 31u), widx = 1;
        uint32_t pp = 0, n = 0, depth = 0, inb = 0, nleaf = 0, bad = 0, done = 0;
        uint32_t stk = 0;  // lane d: id of the branch at depth d on the current path
        while (!done && !bad) {
            if (navail < 10u) {
                const uint32_t nw = widx < 64u ? read_lane(dv0, widx) : read_lane(dv1, min(widx - 64u, 63u));
                win |= (unsigned long long)nw << navail;
                navail += 32u;
                ++widx;
            }
            const uint32_t bits = (uint32_t)win;
            if (!(bits & 1u)) {  // branch
                if (pp + 1u > P || n >= 2u * kNumSym - 1u || depth >= 31u) {  // (depth 32+: deeper than the reference's 32-bit codes)
                    bad = 1;
                    break;
                }
                if (l == 0) d.node[n] = 0u;  // child_b: filled in when its subtree starts
                stk = l == depth ? n : stk;
                inb &= ~(1u << depth);
                ++depth;
                ++n;
                win >>= 1;
                navail -= 1u;
                pp += 1u;
            } else {  // leaf
                const uint32_t sym = (bits >> 1) & 511u;
                if (pp + 10u > P || n >= 2u * kNumSym - 1u || sym > 260u) {
                    bad = 1;
                    break;
                }
                if (l == 0) d.node[n] = kNodeLeaf | sym;
                ++n;
                ++nleaf;
                win >>= 10;
                navail -= 10u;
                pp += 10u;
                // close the branches whose child_b this leaf ends: back to the deepest branch still in its child_a
                const uint32_t open = ~inb & ((depth >= 32u ? 0u : (1u << depth)) - 1u);
                if (!open) {
                    done = 1;  // the root's subtree is complete (a single-leaf tree ends here at once)
                } else {
                    depth = 32u - (uint32_t)__clz((int)open);  // = index of that branch + 1
                    const uint32_t x = read_lane(stk, depth - 1u);
                    if (l == 0) d.node[x] = n;  // its child_b is the node that comes next
                    inb |= 1u << (depth - 1u);
                }
            }
        }
        if (l == 0) {
            d.err = (bad || !done || nleaf > (uint32_t)kNumSym || bit0 + pp > bit_end) ? 1u : 0u;
            d.nnode = n;
            d.nleaf = nleaf;
            d.code0 = bit0 + pp;  // where the codes start
        }
    }
    __syncthreads();
    if (d.err) {  // (block-uniform) no complete tree inside the payload, too many nodes, a symbol out of range, too deep
        if (tid == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
        return;
    }
    {
        const uint32_t nn = d.nnode;
        for (uint32_t e = tid; e < (1u << kLutBits); e += kDecThreads) {
            uint32_t nd = 0, len = 0, wv = d.node[0];
            while (!(wv & kNodeLeaf) && len < kLutBits) {
                nd = min(((e >> len) & 1u) ? wv : nd + 1u, nn - 1u);
                wv = d.node[nd];
                ++len;
            }
            // a single-leaf tree has depth 0: the stream still spends 1 bit per symbol (hzr_decode.c:290,463-470)
            uint32_t ent = tok_entry(wv & 511u, len ? len : 1u);
            if (!(wv & kNodeLeaf)) ent = long_prefix_entry(d, nd);  // longer codes continue from nd
            d.lut[e] = ent;
        }
    }
    __syncthreads();
    } else {
    // RecoverTree (hzr_decode.c:263-333) in three steps.  The description is pre-order: '1' + 9-bit symbol = leaf, '0' = branch
    // followed by child_a then child_b.  Nodes are numbered in pre-order, so child_a = id + 1.
    //   A (one lane, scalar unit): cut the bit string into nodes; keep the count of subtrees still open after each
    //   B (a thread per node): a branch finds the end of its child_a subtree -- the first later node after which one subtree
    //     fewer is open than after the branch itself -- and so its child_b; parent links follow
    //   C (a thread per node): walk up to the root for depth and code; leaves fill the tables, branches at depth 10 mark
    //     where codes longer than the table index continue
    // A, in parallel too: position p of the description starts a token iff it is reached from position 0 by the jumps
    // p -> p + (bit p ? 10 : 1).  Pointer doubling marks every reached position in ten rounds (<= 521 tokens); one block
    // scan over the marks numbers the nodes and counts the leaves, and the subtrees still open after a node follow from the
    // two counts (1 + branches - leaves), so the first node that leaves none open ends the tree.
    constexpr uint32_t kMaxDesc = 11u * kNumSym + 16u;  // description bits (11 S - 1) and a little slack
    // scratch: lut, cend, node and lut2 are contiguous and free until the tree is known -- two jump arrays (a round reads one and
    // writes the other: one barrier per round instead of two) and the marks behind them, inside lut2, which nothing else writes
    // before the second-level tables are filled
    constexpr uint32_t kJumpLen = (kMaxDesc + 2u + 7u) & ~7u;
    uint16_t* jump_a = reinterpret_cast<uint16_t*>(d.lut);       // [kMaxDesc + 1]
    uint16_t* jump_b = jump_a + kJumpLen;
    uint32_t* mark = d.lut + kJumpLen;                            // bit per position (= behind jump_b)
    static_assert(offsetof(DecLds, cend) == offsetof(DecLds, lut) + sizeof(d.lut) && offsetof(DecLds, node) == offsetof(DecLds, cend) + sizeof(d.cend) &&
                      offsetof(DecLds, lut2) == offsetof(DecLds, node) + sizeof(d.node),
                  "the token cut uses lut .. lut2 as one piece of scratch");
    static_assert(kJumpLen * 4u + (kMaxDesc / 32u + 2u) * 4u <= sizeof(d.lut) + sizeof(d.cend) + sizeof(d.node) + sizeof(d.lut2), "token-cut scratch does not fit");
    static_assert(kJumpLen * 4u >= sizeof(d.lut) + sizeof(d.cend) + sizeof(d.node), "the marks must lie behind the node array (written while they are read)");
    const uint32_t P = min(kMaxDesc, bit_end - bit0);
    auto dbit = [&](uint32_t p) -> uint32_t { return (d.stage[(bit0 + p) >> 5] >> ((bit0 + p) & 31u)) & 1u; };
    for (uint32_t pz = tid; pz < kMaxDesc / 32 + 2; pz += kDecThreads) mark[pz] = pz == 0 ? 1u : 0u;  // position 0 is reached
    for (uint32_t pp = tid; pp <= P; pp += kDecThreads) jump_a[pp] = (uint16_t)(pp < P ? min(P, pp + (dbit(pp) ? 10u : 1u)) : P);
    if (tid == 0) {
        d.err = 0;
        d.endpos = 0xFFFFFFFFu;
    }
    __syncthreads();
    // (a mark set in this round by somebody else may or may not be seen in it: either way only reachable positions get marked,
    //  and what a round must see -- the marks of the rounds before -- is behind a barrier)
#pragma unroll 2
    for (uint32_t round = 0; round < 10; ++round) {
        const uint16_t* jr = (round & 1u) ? jump_b : jump_a;
        uint16_t* jw = (round & 1u) ? jump_a : jump_b;
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q) {
            const uint32_t pp = tid + q * kDecThreads;
            const uint32_t j1 = pp <= P ? jr[pp] : P;
            const uint32_t j2 = jr[j1];
            if (pp < P && ((mark[pp >> 5] >> (pp & 31u)) & 1u) && j1 < P) atomicOr(&mark[j1 >> 5], 1u << (j1 & 31u));
            if (pp <= P) jw[pp] = (uint16_t)j2;
        }
        __syncthreads();
    }
    // nodes in position order: thread t owns positions [3t, 3t+3)
    uint32_t mine_n = 0, mine_l = 0, isnode[3], isleaf[3];
#pragma unroll
    for (uint32_t q = 0; q < 3; ++q) {
        const uint32_t pp = 3u * tid + q;
        isnode[q] = pp < P ? (mark[pp >> 5] >> (pp & 31u)) & 1u : 0u;
        isleaf[q] = isnode[q] ? dbit(pp) : 0u;
        mine_n += isnode[q];
        mine_l += isleaf[q];
    }
    const uint32_t packed = mine_n | (mine_l << 16);
    const uint32_t tincl = wave_scan_add(packed);
    if (l == 63) d.wsum[w] = tincl;
    __syncthreads();
    uint32_t tpre = 0;
    for (uint32_t i = 0; i < w; ++i) tpre += d.wsum[i];
    uint32_t run = tpre + tincl - packed;  // nodes | leaves << 16 before this thread's positions
    // (the marks and jumps are dead from here on: the node arrays may overwrite them -- but only after everybody has
    //  read its marks, which the barrier above guarantees)
    uint16_t* t_open = reinterpret_cast<uint16_t*>(d.cend);  // [kNodeSlots]  subtrees still open after the node
    uint16_t* t_par = t_open + kNodeSlots;                     // parent | child_b? << 15
    static_assert(2 * kNodeSlots * sizeof(uint16_t) <= sizeof(d.cend), "tree scratch does not fit");
    uint32_t my_err = 0;
#pragma unroll
    for (uint32_t q = 0; q < 3; ++q) {
        if (isnode[q]) {
            const uint32_t pp = 3u * tid + q;
            const uint32_t id = run & 0xFFFFu, leaves_before = run >> 16;
            const uint32_t open_after = 1u + (id + 1u - (leaves_before + isleaf[q])) - (leaves_before + isleaf[q]);
            if (id < 2u * kNumSym - 1) {
                const uint32_t sym = (peek32(d.stage, bit0 + pp) >> 1) & 511u;
                d.node[id] = isleaf[q] ? (kNodeLeaf | sym) : 0u;
                // (t_open and friends live in cend, the marks in lut + the head of cend: write them after the barrier below)
                if (isleaf[q] && open_after == 0) atomicMin(&d.endpos, pp);
            }
            run += 1u | (isleaf[q] << 16);
        }
    }
    __syncthreads();  // endpos is final; all marks have been consumed
    const uint32_t endpos = d.endpos;
    run = tpre + tincl - packed;
#pragma unroll
    for (uint32_t q = 0; q < 3; ++q) {
        if (isnode[q]) {
            const uint32_t pp = 3u * tid + q;
            const uint32_t id = run & 0xFFFFu, leaves_before = run >> 16;
            if (pp <= endpos && id < kNodeSlots) {
                const uint32_t lv = leaves_before + isleaf[q];
                t_open[id] = (uint16_t)(1u + (id + 1u - lv) - lv);
                if (isleaf[q] && (d.node[id] & 511u) > 260u) my_err = 1;
                if (pp == endpos) {
                    d.nnode = id + 1;
                    d.nleaf = lv;
                    d.code0 = bit0 + pp + 10;  // where the codes start
                }
            }
            run += 1u | (isleaf[q] << 16);
        }
    }
    if (tid == 0) t_par[0] = 0;
    // no complete tree inside the payload, too many nodes, or a symbol out of range
    if (wg_any(d, 1, my_err || endpos == 0xFFFFFFFFu || endpos >= P)) {
        if (tid == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
        return;
    }
    if (d.nnode > 2u * kNumSym - 1 || d.nleaf > (uint32_t)kNumSym || d.code0 > bit_end) {  // (block-uniform)
        if (tid == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
        return;
    }
    for (uint32_t i = tid; i < (1u << kLutBits); i += kDecThreads) d.lut[i] = kLutSlow;  // (steps B / C below fill it)
    const uint32_t nn = d.nnode, nleaf = d.nleaf;
    {  // B (nn <= 521 < kDecThreads: node tid).  Most subtrees are small: a few steps alone; the long searches -- the root's
       // child_a subtree is half the tree -- are then taken one at a time by the whole wave, 64 positions a step.
        const uint32_t i = tid;
        const bool isbr = i < nn && !(d.node[min(i, kNodeSlots - 1u)] & kNodeLeaf);
        const uint32_t want = isbr ? (uint32_t)t_open[i] - 1u : 0u;  // open subtrees once child_a's subtree is read
        uint32_t j = i + 1;
        bool found = !isbr;
        for (int st = 0; st < 6; ++st) {
            if (!any_lane(!found)) break;
            if (!found) {
                if (j + 1 >= nn || (uint32_t)t_open[j] == want)
                    found = true;
                else
                    ++j;
            }
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(!found);
        while (todo) {
            const uint32_t srcl = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            uint32_t jj = read_lane(j, srcl);
            const uint32_t ww = read_lane(want, srcl);
            uint32_t res = jj;
            for (;;) {  // (position nn - 1 always ends the search: a complete description has the match before it)
                const uint32_t pz = jj + l;
                const bool hit = pz + 1 >= nn || (uint32_t)t_open[min(pz, kNodeSlots - 1u)] == ww;
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
                if (hm) {
                    res = jj + (uint32_t)__builtin_ctzll(hm);
                    break;
                }
                jj += 64;
            }
            if (l == srcl) j = res;
        }
        if (isbr) {
            const uint32_t cb = j + 1;  // child_b follows child_a's subtree
            d.node[i] = cb;
            t_par[i + 1] = (uint16_t)i;
            t_par[cb < nn ? cb : i + 1] = (uint16_t)(i | 0x8000u);
        }
    }
    __syncthreads();
    uint32_t deep = 0;
    for (uint32_t i = tid; i < nn; i += kDecThreads) {  // C
        // one walk to the root: the decisions come deepest first, so shifting them in from the right leaves the root's at
        // bit 0 -- the code as the stream has it (root first, LSB first).  (Depths beyond 32 are refused below.)
        uint32_t depth = 0, cur = i, code = 0;
        while (cur != 0 && depth < 40) {
            const uint32_t pw = t_par[cur];
            code = (code << 1) | (pw >> 15);
            cur = pw & 0x7FFFu;
            ++depth;
        }
        const uint32_t w_node = d.node[i];
        if (w_node & kNodeLeaf) {
            if (depth > 31) deep = 1;
            // a single-leaf tree has depth 0: the stream still spends 1 bit per symbol (hzr_decode.c:290,463-470)
            const uint32_t elen = depth ? depth : 1u;
            if (elen <= kLutBits)  // every leaf of <= 10 bits owns the entries code + m * 2^len
                for (uint32_t e = code; e < (1u << kLutBits); e += 1u << elen) d.lut[e] = tok_entry(w_node & 511u, elen);
        } else if (depth == kLutBits) {
            d.lut[code] = long_prefix_entry(d, i);  // codes longer than the table index continue from here
        }
    }
    if (wg_any(d, 2, deep != 0)) {  // deeper than the reference's decoder supports (hzr_decode.c: 32-bit codes)
        if (tid == 0) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
        return;
    }
    }
    {  // second-level tables (the barrier behind either tree recovery has published the slots): thread = slot * 32 + the next 5 bits
        const uint32_t sl = tid >> kSubBits, nsl = min(d.nslot, kSubSlots);
        if (nsl) {  // (block-uniform)
            if (sl < nsl) {
                uint32_t nd = min(d.slot_node[sl], kNodeSlots - 1u), len = 0, wv = d.node[nd];
                while (!(wv & kNodeLeaf) && len < kSubBits) {
                    nd = min(((tid >> len) & 1u) ? wv : nd + 1u, kNodeSlots - 1u);
                    wv = d.node[nd];
                    ++len;
                }
                d.lut2[tid] = (wv & kNodeLeaf) ? tok_entry(wv & 511u, kLutBits + len) : (kLutLong | nd);
            }
            __syncthreads();
        }
    }
    const uint32_t code0 = d.code0;
    DEC_STAMP(2);
    DEC_STAMP(3);

    // ---- chunks ---------------------------------------------------------------
    const uint32_t nbits = bit_end - code0;
    uint32_t nchunk = (nbits + kDecMinChunkBits - 1) / kDecMinChunkBits;
    nchunk = nchunk < 1 ? 1u : nchunk > kDecThreads ? kDecThreads : nchunk;
    const uint32_t S = (nbits + nchunk - 1) / nchunk;  // bits per chunk
    const bool mine = tid < nchunk;
    const uint32_t limit = tid + 1 == nchunk ? bit_end : min(bit_end, code0 + (tid + 1) * S);
    uint32_t start = code0 + tid * S, produced = 0, spec_err = 0;
    if (mine) {
        d.cend[tid] = dec_chunk<kDecScan>(d, start, limit, bit_end, produced, nullptr, 0, 0, spec_err);  // boundaries only
    }
#if defined(RSPT_DIAG) && defined(RSPT_DEC_ONEROUND)  // timing probe, diagnostic builds only
    for (uint32_t round = 0; round < 1; ++round) {
#else
    for (uint32_t round = 0; round < nchunk; ++round) {  // (bounded: the first wrong chunk is right after every round)
#endif
        __syncthreads();  // the previous round's flag has been read by everybody
        if (tid == 0) d.changed = 0;
        uint32_t want = start;
        if (mine && tid > 0) want = d.cend[tid - 1];
        __syncthreads();  // everybody has read its predecessor's end before anybody rewrites its own
        // round 0 counts every chunk's bytes (the pre-pass found boundaries only); later rounds redo the chunks that moved
        if (mine && (round == 0 || want != start)) {
            if (want != start) d.changed = 1;
            start = want;
            spec_err = 0;
            produced = 0;
            d.cend[tid] = start >= limit ? start : dec_chunk<kDecCount>(d, start, limit, bit_end, produced, nullptr, 0, 0, spec_err);
        }
        __syncthreads();
        if (!d.changed) break;
    }
    DEC_STAMP(4);
    // Every chunk now starts where its predecessor ended: the starts are the true code boundaries.  The chunk that holds the
    // end of the data may have decoded the final byte's pad bits as codes, and the chunks behind it decode nothing real:
    // the final pass is therefore also bounded by the byte count, as the reference's decoder is (hzr_decode.c:463-567).
    const uint32_t mycount = mine ? produced : 0u;
    const uint32_t incl = wave_scan_add(mycount);
    if (l == 63) d.wsum[w] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t i = 0; i < w; ++i) pre += d.wsum[i];
    const uint32_t o0 = pre + incl - mycount;  // exact for every chunk up to the one that holds the end
    uint32_t e2 = 0, p2 = 0;
#if defined(RSPT_DIAG) && defined(RSPT_DEC_NOWRITE)  // timing probe, diagnostic builds only
    if (mine && start < limit && o0 < out_size) dec_chunk<kDecCount>(d, start, limit, bit_end, p2, out, o0, out_size, e2, out_size - o0);
#else
    uint32_t room = out_size - o0;  // (in a register of its own: rebuilt per token from the spilled scalar otherwise)
    asm volatile("" : "+v"(room));
    if (mine && start < limit && o0 < out_size) dec_chunk<kDecWrite>(d, start, limit, bit_end, p2, out, o0, out_size, e2, room);
#endif
    // sound iff no bad code was met and exactly out_size bytes came out
    const uint32_t inc2 = wave_scan_add(p2);
    __syncthreads();  // (wsum is reused)
    DEC_STAMP(5);
    if (l == 63) d.wsum[w] = inc2;
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t i = 0; i < kDecThreads / 64; ++i) total += d.wsum[i];
    // (a bad code is reported by the lane that met it, a wrong byte count by thread 0: both rare, no vote needed)
    if (e2 || (tid == 0 && total != out_size)) atomicOr((unsigned long long*)&consumed[b], (unsigned long long)kBadBit);
    DEC_STAMP(6);
    if (stamps && tid == 0 && hb < 16384u) stamps[65536u + 2u * hb + 1u] = __builtin_amdgcn_s_memrealtime();
#undef DEC_STAMP
}

// Persistent grid over the hzr blocks: the first block of a workgroup is static, the rest come from a counter.  The order is
// plane-fastest -- dense plane-0 blocks and the light ones of the planes above take turns -- so that a CU's two workgroups are
// in different kinds of block, and in different phases, most of the time (all dense blocks first, the light ones as tail filler:
// 1.729 ms per 64-block batch; in turns: 1.677).
__global__ __launch_bounds__(kDecThreads, 8) void k_dec_block(const uint8_t* __restrict__ src, uint64_t src_stride, Geom g,
                                                          const uint32_t* __restrict__ dec_nb, const uint64_t* __restrict__ blk_off,
                                                          uint8_t* __restrict__ planes, uint64_t* __restrict__ consumed,
                                                          unsigned long long* __restrict__ stamps, const CrcConsts* __restrict__ vcc,
                                                          const uint64_t* __restrict__ pidx, uint32_t* __restrict__ counter, uint32_t total) {
    __shared__ uint32_t s_next;
    for (uint32_t pass = 0;; ++pass) {
        __syncthreads();  // everyone is done with the previous block's LDS (and with s_next)
        if (threadIdx.x == 0) s_next = pass == 0 ? blockIdx.x : gridDim.x + atomicAdd(counter, 1u);
        __syncthreads();
        const uint32_t i = s_next;
        if (i >= total) break;
        const uint32_t k = i % kMaxPlanes, x = i / kMaxPlanes;
        dec_block(k, x % g.nblk, x / g.nblk, src, src_stride, g, dec_nb, blk_off, planes, consumed, stamps, vcc, pidx);
    }
}

// ---------------------------------------------------------------------------
// inverse transform: tiled scans over the flat array.  Tile = 4096 elements,
// 256 threads x 16 consecutive elements.
// ---------------------------------------------------------------------------
constexpr uint32_t kInvTile = 4096;

// v[i0..i0+16) assembled from nb planes, sign-extended from nb bytes (base.cpp:121-138)
__device__ __forceinline__ void load_v16(const uint8_t* planes, const Geom& g, uint32_t b, uint32_t nb, uint32_t i0, uint32_t cnt, uint32_t v[16]) {
    uint32_t pw[4][4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        uint4 w = make_uint4(0, 0, 0, 0);
        if (k < nb && cnt) w = *reinterpret_cast<const uint4*>(planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + i0);
        pw[k][0] = w.x;
        pw[k][1] = w.y;
        pw[k][2] = w.z;
        pw[k][3] = w.w;
    }
    // 4 x 4 byte transposes (the mirror image of the front end's: preprocess.hip, transform_item): two v_perm per sample, then
    // the sign of an nb-byte value (planes >= nb were loaded as zero)
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t t01 = __builtin_amdgcn_perm(pw[1][q], pw[0][q], 0x05010400u), u01 = __builtin_amdgcn_perm(pw[1][q], pw[0][q], 0x07030602u);
        const uint32_t t23 = __builtin_amdgcn_perm(pw[3][q], pw[2][q], 0x05010400u), u23 = __builtin_amdgcn_perm(pw[3][q], pw[2][q], 0x07030602u);
        v[4 * q] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
        v[4 * q + 1] = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
        v[4 * q + 2] = __builtin_amdgcn_perm(u23, u01, 0x05040100u);
        v[4 * q + 3] = __builtin_amdgcn_perm(u23, u01, 0x07060302u);
    }
    if (nb < 4) {  // (uniform)
        const uint32_t width = 8 * nb;
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) v[e] = (uint32_t)__builtin_amdgcn_sbfe((int32_t)v[e], 0, width);
    }
    if (cnt < 16) {
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) v[e] = e < cnt ? v[e] : 0u;
    }
}

// block-wide exclusive scan of one value per thread (256 threads); also returns the total
template <bool XOR>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t& total) {
    const uint32_t l = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        uint32_t o = (uint32_t)__shfl_up((int)inc, dd, 64);
        if (l >= (uint32_t)dd) inc = XOR ? (inc ^ o) : (inc + o);
    }
    if (l == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (uint32_t i = 0; i < 4; ++i) {
        if (i < w) pre = XOR ? (pre ^ s_w[i]) : (pre + s_w[i]);
        tot = XOR ? (tot ^ s_w[i]) : (tot + s_w[i]);
    }
    total = tot;
    __syncthreads();
    const uint32_t incl_before = XOR ? (inc ^ v) : (inc - v);
    return XOR ? (pre ^ incl_before) : (pre + incl_before);
}

// PASS 0: tile XOR totals.  PASS 1: tile sums of o+128 (needs XOR carries).
// PASS 2: final values p -> planar.  XDELTA=false: p = v, single pass.
template <int PASS, bool XDELTA>
__global__ __launch_bounds__(256) void k_inv_tile(const uint8_t* __restrict__ planes, Geom g, const uint32_t* __restrict__ dec_nb,
                                                 uint32_t ntile, uint32_t* __restrict__ txor, uint32_t* __restrict__ tsum,
                                                 int32_t* __restrict__ planar) {
    __shared__ uint32_t s_w[4];
    const uint32_t tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const uint32_t nb = dec_nb[b];  // planes of THIS stream (k_dec_frame)
    const uint32_t i0 = tile * kInvTile + tid * 16;
    const uint32_t cnt = i0 < g.N ? min(16u, g.N - i0) : 0u;
    uint32_t v[16];
    load_v16(planes, g, b, nb, i0, cnt, v);
    uint32_t p[16];
    if (XDELTA) {
        uint32_t x = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) x ^= v[e];
        uint32_t tot;
        const uint32_t xpre = block_excl_scan<true>(x, s_w, tot);
        if (PASS == 0) {
            if (tid == 0) txor[(size_t)b * ntile + tile] = tot;
            return;
        }
        uint32_t o = txor[(size_t)b * ntile + tile] ^ xpre;  // xor_decode_32: o[i] = o[i-1]^v[i] (utils.cpp:232-236)
        uint32_t dsum = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o ^= v[e];
            p[e] = o + 128u;  // offset_32(+128)
            if ((uint32_t)e < cnt) dsum += p[e];
        }
        const uint32_t spre = block_excl_scan<false>(dsum, s_w, tot);
        if (PASS == 1) {
            if (tid == 0) tsum[(size_t)b * ntile + tile] = tot;
            return;
        }
        uint32_t acc = tsum[(size_t)b * ntile + tile] + spre;  // delta_decode: p[i] = p[i-1] + d[i] (utils.cpp:204-213)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc += p[e];
            p[e] = acc;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) p[e] = v[e];
    }
    // Each thread holds 16 consecutive results: stored directly, a wave would write 16-byte pieces 64 bytes apart.  Whole
    // tiles go through LDS instead (rows of 17 words: conflict-free both ways) and leave as 1 KiB per wave-store.
    __shared__ uint32_t s_t[256 * 17];
    const uint32_t tile0 = tile * kInvTile;
    const bool whole = tile0 + kInvTile <= g.N && (((size_t)b * g.N + tile0) & 3u) == 0;  // (block-uniform)
    if (whole) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s_t[tid * 17 + e] = p[e];
        __syncthreads();
        uint4* o4 = reinterpret_cast<uint4*>(planar + (size_t)b * g.N + tile0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = 4u * (tid + 256u * q);  // element of the tile
            const uint32_t* r = s_t + (j >> 4) * 17 + (j & 15u);
            o4[tid + 256u * q] = make_uint4(r[0], r[1], r[2], r[3]);
        }
        return;
    }
    int32_t* dstp = planar + (size_t)b * g.N + i0;
    for (uint32_t e = 0; e < cnt; ++e) dstp[e] = (int32_t)p[e];
}

// exclusive scan of the per-tile totals of one block, in place (one workgroup per block)
template <bool XOR>
__global__ __launch_bounds__(1024) void k_inv_scan_tiles(uint32_t* __restrict__ t, uint32_t ntile) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    uint32_t* a = t + (size_t)blockIdx.x * ntile;
    const uint32_t tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ntile; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < ntile ? a[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            uint32_t o = (uint32_t)__shfl_up((int)inc, dd, 64);
            if (l >= (uint32_t)dd) inc = XOR ? (inc ^ o) : (inc + o);
        }
        if (l == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t pre = s_carry, tot = s_carry;
        for (uint32_t q = 0; q < 16; ++q) {
            if (q < w) pre = XOR ? (pre ^ s_w[q]) : (pre + s_w[q]);
            tot = XOR ? (tot ^ s_w[q]) : (tot + s_w[q]);
        }
        if (i < ntile) a[i] = XOR ? (pre ^ inc ^ v) : (pre + inc - v);
        __syncthreads();
        if (tid == 0) s_carry = tot;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// The int32 fast path (ns % 256 == 0, nch % 4 == 0): the last pass writes the interleaved samples itself -- no planar
// int32 round trip (1.07 GB written and read again per 64-block batch of the BASELINE shape).  The scan tile is a ROW of
// sixteen lanes x 16 consecutive samples = 256 samples of one channel, scanned with row-local DPP steps; a workgroup takes
// CG channels x (1024 / CG) * 16 samples, so that what it writes is whole sample rows of CG channels (256 bytes for 64).
// ---------------------------------------------------------------------------
constexpr uint32_t kRowTile = 256;

// One pass over the planes gives both carries.  The XOR total of a row tile is plain.  Its SUM depends on the XOR carry c
// that enters it -- sum_i ((c ^ y_i) + 128) with y_i the tile-local XOR prefixes -- but only through the number of y_i
// that have each bit set: with cnt_b of the 256 values holding bit b,
//     sum_i (c ^ y_i) = sum_b 2^b (c_b ? 256 - cnt_b : cnt_b)          (mod 2^32).
// The counts are kept bit-sliced: plane k holds bit k of every cnt_b (nine planes for counts up to 256), so that
//     sum_b 2^b cnt_b [over the bits b in a mask m] = sum_k 2^k (P_k & m)    -- an integer sum of masked plane words.
// k_inv_rows leaves (XOR total, P_0..P_8) per row tile; k_inv_scan_rows turns them into the two exclusive carries.
constexpr uint32_t kRowPlanes = 9;
constexpr uint32_t kRowRec = 1 + kRowPlanes;  // words per row tile record

__device__ __forceinline__ void full_add(uint32_t a, uint32_t b, uint32_t c, uint32_t& s, uint32_t& cy) {
    const uint32_t x = a ^ b;
    s = x ^ c;
    cy = (x & c) | (~x & a);  // majority (one v_bfi)
}

__global__ __launch_bounds__(256) void k_inv_rows(const uint8_t* __restrict__ planes, Geom g, const uint32_t* __restrict__ dec_nb, uint32_t nrow,
                                                 uint32_t* __restrict__ rec) {
    const uint32_t b = blockIdx.y, tid = threadIdx.x;
    const uint32_t i0 = (blockIdx.x * 256u + tid) * 16u;
    if (i0 >= g.N) return;  // (whole rows leave together)
    uint32_t v[16];
    load_v16(planes, g, b, dec_nb[b], i0, 16u, v);
    uint32_t x = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) x ^= v[e];
    const uint32_t xinc = row_scan_prefix(x, 0u, [](uint32_t a, uint32_t c) { return a ^ c; });
    uint32_t y[16], o = xinc ^ x;  // XOR of the lanes below in the row
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        o ^= v[e];
        y[e] = o;
    }
    // sixteen one-bit columns -> a 5-bit count per bit position (carry-save adders)
    uint32_t P[kRowPlanes];
    {
        uint32_t s0, c0, s1, c1, s2, c2, s3, c3, s4, c4, t0, d0, t1, d1, u0, f0, u1, f1, u2, f2, h0, i0_;
        full_add(y[0], y[1], y[2], s0, c0);
        full_add(y[3], y[4], y[5], s1, c1);
        full_add(y[6], y[7], y[8], s2, c2);
        full_add(y[9], y[10], y[11], s3, c3);
        full_add(y[12], y[13], y[14], s4, c4);
        full_add(s0, s1, s2, t0, d0);
        full_add(s3, s4, y[15], t1, d1);
        P[0] = t0 ^ t1;
        const uint32_t e0 = t0 & t1;
        full_add(c0, c1, c2, u0, f0);
        full_add(c3, c4, d0, u1, f1);
        full_add(d1, e0, u0, u2, f2);
        P[1] = u1 ^ u2;
        const uint32_t g0 = u1 & u2;
        full_add(f0, f1, f2, h0, i0_);
        P[2] = h0 ^ g0;
        const uint32_t j0 = h0 & g0;
        P[3] = i0_ ^ j0;
        P[4] = i0_ & j0;
    }
    // ... summed over the sixteen lanes of the row (ripple adders on the planes; lane 15 ends with the total)
#define RSPT_ROW_ADD(CTRL, NP)                                  \
    {                                                           \
        uint32_t cy = 0;                                        \
        _Pragma("unroll") for (int k = 0; k < NP; ++k) {        \
            const uint32_t other = dpp<CTRL>(0u, P[k]);         \
            uint32_t sum;                                       \
            full_add(P[k], other, cy, sum, cy);                 \
            P[k] = sum;                                         \
        }                                                       \
        P[NP] = cy;                                             \
    }
    RSPT_ROW_ADD(0x111, 5)
    RSPT_ROW_ADD(0x112, 6)
    RSPT_ROW_ADD(0x114, 7)
    RSPT_ROW_ADD(0x118, 8)
#undef RSPT_ROW_ADD
    if ((tid & 15u) == 15u) {
        uint32_t* r = rec + ((size_t)b * nrow + (i0 >> 8)) * kRowRec;
        r[0] = xinc;
#pragma unroll
        for (uint32_t k = 0; k < kRowPlanes; ++k) r[1 + k] = P[k];
    }
}

// workgroup-wide inclusive scan of one value per thread (1024 threads), on top of a running carry
template <bool XOR>
__device__ __forceinline__ uint32_t wg_scan_incl(uint32_t v, uint32_t* s_w, uint32_t& carry) {
    const uint32_t l = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = XOR ? wave_scan_incl(v, 0u, [](uint32_t a, uint32_t c) { return a ^ c; })
                       : wave_scan_incl(v, 0u, [](uint32_t a, uint32_t c) { return a + c; });
    __syncthreads();  // (s_w is reused)
    if (l == 63u) s_w[w] = inc;
    __syncthreads();
    uint32_t pre = carry, tot = carry;
    for (uint32_t q = 0; q < 16; ++q) {
        if (q < w) pre = XOR ? (pre ^ s_w[q]) : (pre + s_w[q]);
        tot = XOR ? (tot ^ s_w[q]) : (tot + s_w[q]);
    }
    carry = tot;
    return XOR ? (pre ^ inc) : (pre + inc);
}

// row tile records of one block -> exclusive XOR carry and exclusive sum carry of every row tile (one workgroup per block)
__global__ __launch_bounds__(1024) void k_inv_scan_rows(const uint32_t* __restrict__ rec, uint32_t nrow, uint32_t* __restrict__ txor,
                                                       uint32_t* __restrict__ tsum) {
    __shared__ uint32_t s_w[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t* r0 = rec + (size_t)blockIdx.x * nrow * kRowRec;
    uint32_t* ox = txor + (size_t)blockIdx.x * nrow;
    uint32_t* os = tsum + (size_t)blockIdx.x * nrow;
    uint32_t carry_x = 0, carry_s = 0;
    for (uint32_t base = 0; base < nrow; base += 1024) {
        const uint32_t i = base + tid;
        const bool in = i < nrow;
        const uint32_t* r = r0 + (size_t)i * kRowRec;
        const uint32_t x = in ? r[0] : 0u;
        const uint32_t c = wg_scan_incl<true>(x, s_w, carry_x) ^ x;  // the XOR carry that enters row tile i
        uint32_t sum = 0;
        if (in) {
            sum = kRowTile * c + kRowTile * 128u;
#pragma unroll
            for (uint32_t k = 0; k < kRowPlanes; ++k) {
                const uint32_t pk = r[1 + k];
                sum += ((pk & ~c) - (pk & c)) << k;
            }
        }
        const uint32_t sinc = wg_scan_incl<false>(sum, s_w, carry_s);
        if (in) {
            ox[i] = c;
            os[i] = sinc - sum;
        }
    }
}

// element s of a channel's piece sits in column inv_col(s) of its LDS row: the sixteen words of a thread are rotated by
// (thread / 4) so that the lanes of a wave, whose spans start 16 words apart, do not meet in four banks
__device__ __forceinline__ uint32_t inv_col(uint32_t s) { return (s & ~15u) | ((s + (s >> 6)) & 15u); }

template <bool XDELTA, int CG>
__global__ __launch_bounds__(1024) void k_inv_native(const uint8_t* __restrict__ planes, Geom g, const uint32_t* __restrict__ dec_nb, uint32_t nrow,
                                                    const uint32_t* __restrict__ txor, const uint32_t* __restrict__ tsum, uint8_t* __restrict__ dst) {
    constexpr uint32_t TPC = 1024u / CG, S = TPC * 16u, ROW = S + 1u;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t* tile = reinterpret_cast<uint32_t*>(lds);  // [CG][ROW]
    const uint32_t tid = threadIdx.x, c = tid / TPC, q = tid % TPC;
    const uint32_t b = blockIdx.z, cg0 = blockIdx.y * CG, s0 = blockIdx.x * S;
    const uint32_t ncg = min((uint32_t)CG, g.nch - cg0);
    const uint32_t Sn = min(S, g.ns - s0);  // a multiple of 256
    if (c < ncg && q * 16u < Sn) {  // (row-uniform)
        const uint32_t i0 = (cg0 + c) * g.ns + s0 + q * 16u;
        uint32_t v[16], p[16];
        load_v16(planes, g, b, dec_nb[b], i0, 16u, v);
        if (XDELTA) {
            uint32_t x = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) x ^= v[e];
            const uint32_t xinc = row_scan_prefix(x, 0u, [](uint32_t a, uint32_t d) { return a ^ d; });
            const size_t row = (size_t)b * nrow + (i0 >> 8);
            uint32_t o = txor[row] ^ xinc ^ x, dsum = 0;  // xor_decode_32 (utils.cpp:232-236), offset_32(+128)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                o ^= v[e];
                p[e] = o + 128u;
                dsum += p[e];
            }
            const uint32_t sinc = row_scan_prefix(dsum, 0u, [](uint32_t a, uint32_t d) { return a + d; });
            uint32_t acc = tsum[row] + sinc - dsum;  // delta_decode (utils.cpp:204-213)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc += p[e];
                p[e] = acc;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) p[e] = v[e];
        }
        uint32_t* r = tile + c * ROW + q * 16u;
        const uint32_t rot = q >> 2;
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) r[(e + rot) & 15u] = p[e];
    }
    __syncthreads();
    // sample rows out: 16-byte pieces (sample t, channels 4 c4 .. 4 c4 + 3), channel fastest
    const uint32_t cpr = ncg >> 2;
    const uint32_t total = Sn * cpr;
    uint8_t* o = dst + (size_t)b * g.block_bytes;
    for (uint32_t u = tid; u < total; u += 1024u) {
        const uint32_t t = u / cpr, c4 = u - t * cpr;
        const uint32_t* r = tile + (4u * c4) * ROW + inv_col(t);
        uint32_t x0 = r[0], x1 = r[ROW], x2 = r[2 * ROW], x3 = r[3 * ROW];
        if (g.be) {  // big-endian samples out (rspt_hip_set_byte_order): one v_perm per sample here instead of a pass of its own over the block
            x0 = __builtin_amdgcn_perm(x0, x0, 0x00010203u);
            x1 = __builtin_amdgcn_perm(x1, x1, 0x00010203u);
            x2 = __builtin_amdgcn_perm(x2, x2, 0x00010203u);
            x3 = __builtin_amdgcn_perm(x3, x3, 0x00010203u);
        }
        *reinterpret_cast<uint4*>(o + ((size_t)(s0 + t) * g.nch + cg0 + 4u * c4) * 4u) = make_uint4(x0, x1, x2, x3);
    }
}

// [nch][ns] int32 -> interleaved native bytes (convert_i32_to_native, utils.cpp:51-121, LE branches)
template <int BPS>
__global__ __launch_bounds__(256) void k_planar_native(const int32_t* __restrict__ planar, Geom g, uint32_t T, uint8_t* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    int32_t* tile = reinterpret_cast<int32_t*>(lds);  // [nch][T+1]
    const uint32_t tid = threadIdx.x, b = blockIdx.y;
    const uint32_t s0 = blockIdx.x * T;
    const uint32_t Tn = min(T, g.ns - s0);
    const uint32_t RS = T + 1;
    const uint32_t total = g.nch * Tn;
    for (uint32_t q = tid; q < total; q += 256) {
        const uint32_t c = q / Tn, t = q - c * Tn;
        tile[c * RS + t] = planar[(size_t)b * g.N + (size_t)c * g.ns + s0 + t];
    }
    __syncthreads();
    uint8_t* o = dst + (size_t)b * g.block_bytes + (size_t)s0 * g.nch * BPS;
    const bool al4 = (BPS == 4) && ((reinterpret_cast<uintptr_t>(o) & 3u) == 0);
    for (uint32_t q = tid; q < total; q += 256) {
        const uint32_t t = q / g.nch, c = q - t * g.nch;
        uint32_t v = (uint32_t)tile[c * RS + t];
        if (BPS > 1 && g.be) v = __builtin_amdgcn_perm(v, v, 0x00010203u) >> (8 * (4 - BPS));  // big-endian samples out: the low BPS bytes reversed
        uint8_t* p = o + (size_t)q * BPS;
        if (al4) {
            *reinterpret_cast<uint32_t*>(p) = v;
        } else {
#pragma unroll
            for (int k = 0; k < BPS; ++k) p[k] = (uint8_t)(v >> (8 * k));
        }
    }
}

// The common shape -- int32 samples, nch % 4 == 0, ns % 4 == 0, 16-byte aligned output -- with 16-byte global accesses and
// no division per element: a tile is T4 samples x nch channels; rows are read four samples at a time, columns written
// four channels at a time (LDS rows of T4 + 1 words: the transposed reads are conflict-free).
__global__ __launch_bounds__(256) void k_planar_native_i32x4(const int32_t* __restrict__ planar, Geom g, uint32_t T4, uint8_t* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    int32_t* tile = reinterpret_cast<int32_t*>(lds);  // [nch][T4+1]
    const uint32_t tid = threadIdx.x, b = blockIdx.y;
    const uint32_t s0 = blockIdx.x * T4;
    const uint32_t Tn = min(T4, g.ns - s0);  // a multiple of 4
    const uint32_t RS = T4 + 1;
    const uint32_t qpr = Tn >> 2;  // 16-byte pieces per channel row
    {
        // piece u = c * qpr + t4; a thread's pieces are 256 apart: (c, t4) advance by a fixed step
        const uint32_t step_c = 256u / qpr, step_t = 256u - step_c * qpr;
        uint32_t c = tid / qpr, t4 = tid - c * qpr;
        for (uint32_t u = tid; u < g.nch * qpr; u += 256) {
            const int4 v = *reinterpret_cast<const int4*>(planar + (size_t)b * g.N + (size_t)c * g.ns + s0 + 4u * t4);
            int32_t* r = tile + c * RS + 4u * t4;
            r[0] = v.x;
            r[1] = v.y;
            r[2] = v.z;
            r[3] = v.w;
            t4 += step_t;
            const uint32_t carry = t4 >= qpr ? 1u : 0u;
            t4 -= carry ? qpr : 0u;
            c += step_c + carry;
        }
    }
    __syncthreads();
    {
        const uint32_t cpr = g.nch >> 2;  // 16-byte pieces per sample row
        const uint32_t step_t = 256u / cpr, step_c = 256u - step_t * cpr;
        uint32_t t = tid / cpr, c4 = tid - t * cpr;
        int4* o = reinterpret_cast<int4*>(dst + (size_t)b * g.block_bytes + (size_t)s0 * g.nch * 4u);
        for (uint32_t u = tid; u < Tn * cpr; u += 256) {
            const int32_t* r = tile + (4u * c4) * RS + t;
            uint32_t x0 = (uint32_t)r[0], x1 = (uint32_t)r[RS], x2 = (uint32_t)r[2 * RS], x3 = (uint32_t)r[3 * RS];
            if (g.be) {  // big-endian samples out
                x0 = __builtin_amdgcn_perm(x0, x0, 0x00010203u);
                x1 = __builtin_amdgcn_perm(x1, x1, 0x00010203u);
                x2 = __builtin_amdgcn_perm(x2, x2, 0x00010203u);
                x3 = __builtin_amdgcn_perm(x3, x3, 0x00010203u);
            }
            o[u] = make_int4((int)x0, (int)x1, (int)x2, (int)x3);
            c4 += step_c;
            const uint32_t carry = c4 >= cpr ? 1u : 0u;
            c4 -= carry ? cpr : 0u;
            t += step_t + carry;
        }
    }
}

}  // namespace rspt
