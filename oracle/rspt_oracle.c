/*
 * rspt_oracle.c -- CPU restatement of the rspt signal_packer hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see rspt_oracle.h).  Written from the stream
 * grammar and the observable behaviour of the reference, not from its text;
 * each function cites the reference file:line it follows (paths under
 * /root/reference/lib_rspt/).  Pinned bit-for-bit against the compiled
 * reference (oracle/_ref) by tests/test_oracle_vs_ref.py and against the
 * committed fixtures by tests/test_oracle_golden.py.
 */
#include "rspt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* hzr constants (lib_hzr/hzr_internal.h:80-121)                             */
/* ------------------------------------------------------------------------ */
#define HZR_BLOCK 65536u    /* max decoded bytes per block   :109 */
#define HZR_NSYM 261        /* 256 literals + 5 zero-run symbols :114 */
#define HZR_RUN_CAP 16662u  /* longest run one token can carry :121 */
#define HZR_MODE_COPY 0
#define HZR_MODE_HUFF 1
#define HZR_MODE_FILL 2

/* zero-run symbol classes: first run length, number of extra bits (:117-121) */
static const unsigned kRunBase[5] = {2, 3, 7, 23, 279};
static const unsigned kRunExtra[5] = {0, 2, 4, 8, 14};

/* ------------------------------------------------------------------------ */
/* CRC-32C (lib_hzr/hzr_crc32c.c:77-97; the LUT path is what the project runs)*/
/* ------------------------------------------------------------------------ */
static uint32_t g_crc_tab[256];
static int g_crc_ready = 0;

static void crc_init(void) {
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t r = b;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (0x82F63B78u & (0u - (r & 1u)));
        g_crc_tab[b] = r;
    }
    g_crc_ready = 1;
}

uint32_t orc_crc32c(const void* data, size_t len) {
    if (!g_crc_ready) crc_init();
    const uint8_t* p = (const uint8_t*)data;
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < len; ++i) c = (c >> 8) ^ g_crc_tab[(c ^ p[i]) & 0xFFu];
    return ~c;
}

uint32_t orc_fnv1a(const void* data, size_t len) {
    const uint8_t* p = (const uint8_t*)data;
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < len; ++i) {
        h ^= p[i];
        h *= 16777619u;
    }
    return h;
}

/* ------------------------------------------------------------------------ */
/* LSB-first bit sink (hzr_encode.c:63-113)                                  */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint8_t* buf;
    size_t nbits;
} bitsink;

static void sink_put(bitsink* s, uint32_t value, unsigned width) {
    /* bit i of `value` lands at stream bit position nbits+i; stream bit q is
     * bit (q&7) of byte q>>3.  The buffer must be zero-initialised and
     * `value` must have no bits at or above `width` (width <= 32). */
    size_t byte = s->nbits >> 3;
    uint64_t v = (uint64_t)value << (s->nbits & 7u);
    for (; v; v >>= 8, ++byte) s->buf[byte] |= (uint8_t)v;
    s->nbits += width;
}

/* ------------------------------------------------------------------------ */
/* tokenizer (hzr_encode.c:133-173 and the identical walk at :410-457)       */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint16_t sym;   /* 0..260 */
    uint16_t extra; /* value of the extra bits (run length - class base) */
} token;

static unsigned run_class(size_t z) { /* z >= 2 */
    if (z == 2) return 0;
    if (z <= 6) return 1;
    if (z <= 22) return 2;
    if (z <= 278) return 3;
    return 4;
}

static size_t tokenize(const uint8_t* in, size_t n, token* out) {
    size_t nt = 0, i = 0;
    while (i < n) {
        if (in[i] != 0) {
            out[nt].sym = in[i];
            out[nt].extra = 0;
            ++nt;
            ++i;
            continue;
        }
        size_t z = 1; /* greedy, capped at 16662 and at the block end (:149) */
        while (z < HZR_RUN_CAP && i + z < n && in[i + z] == 0) ++z;
        if (z == 1) {
            out[nt].sym = 0;
            out[nt].extra = 0;
        } else {
            unsigned c = run_class(z);
            out[nt].sym = (uint16_t)(256 + c);
            out[nt].extra = (uint16_t)(z - kRunBase[c]);
        }
        ++nt;
        i += z;
    }
    return nt;
}

/* ------------------------------------------------------------------------ */
/* Huffman tree (hzr_encode.c:222-283 MakeTree, :177-219 StoreTree)          */
/*                                                                          */
/* The reference repeatedly scans all nodes with a `<=` comparison, which    */
/* selects the 1st and 2nd minimum under the strict order (count ascending,  */
/* node index DESCENDING).  A min-heap on key = count*1024 + (1023-index)    */
/* reproduces that order exactly.                                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t sym; /* >=0 leaf, -1 branch */
    uint32_t count;
    int16_t a, b; /* children (node indices) */
} hnode;

typedef struct {
    uint32_t code[HZR_NSYM];
    uint8_t len[HZR_NSYM];
} codebook;

static void heap_push(uint32_t* h, int* n, uint32_t key) {
    int i = (*n)++;
    h[i] = key;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (h[p] <= h[i]) break;
        uint32_t t = h[p];
        h[p] = h[i];
        h[i] = t;
        i = p;
    }
}

static uint32_t heap_pop(uint32_t* h, int* n) {
    uint32_t top = h[0];
    h[0] = h[--(*n)];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && h[l] < h[m]) m = l;
        if (r < *n && h[r] < h[m]) m = r;
        if (m == i) break;
        uint32_t t = h[m];
        h[m] = h[i];
        h[i] = t;
        i = m;
    }
    return top;
}

/* Builds the tree for `hist`, fills `cb`, and (if sink != NULL) appends the
 * pre-order tree description.  Returns the description length in bits. */
static size_t build_tree(const uint32_t hist[HZR_NSYM], codebook* cb, bitsink* sink) {
    hnode nodes[2 * HZR_NSYM];
    uint32_t heap[2 * HZR_NSYM];
    int nheap = 0, nnodes = 0;
    memset(cb, 0, sizeof(*cb));
    for (int s = 0; s < HZR_NSYM; ++s) { /* leaves in ascending symbol order (:226-234) */
        if (hist[s] == 0) continue;
        nodes[nnodes].sym = s;
        nodes[nnodes].count = hist[s];
        nodes[nnodes].a = nodes[nnodes].b = -1;
        heap_push(heap, &nheap, (hist[s] << 10) | (uint32_t)(1023 - nnodes));
        ++nnodes;
    }
    if (nnodes == 0) return 0; /* :238-240 */
    int root = 0;
    if (nnodes == 1) {
        /* single symbol: a lone leaf with a 1-bit code 0 (:279-282) */
        cb->code[nodes[0].sym] = 0;
        cb->len[nodes[0].sym] = 1;
        if (sink) {
            sink_put(sink, 1, 1);
            sink_put(sink, (uint32_t)nodes[0].sym, 9);
        }
        return 10;
    }
    while (nheap > 1) {
        uint32_t k1 = heap_pop(heap, &nheap);
        uint32_t k2 = heap_pop(heap, &nheap);
        int i1 = 1023 - (int)(k1 & 1023u), i2 = 1023 - (int)(k2 & 1023u);
        nodes[nnodes].sym = -1;
        nodes[nnodes].a = (int16_t)i1; /* lightest -> child_a -> code bit 0 (:263-271) */
        nodes[nnodes].b = (int16_t)i2;
        nodes[nnodes].count = nodes[i1].count + nodes[i2].count;
        heap_push(heap, &nheap, (nodes[nnodes].count << 10) | (uint32_t)(1023 - nnodes));
        root = nnodes++;
    }
    /* pre-order walk; code bit at position `depth` = 1 for child_b (:215-218) */
    struct {
        int16_t node;
        uint8_t depth;
        uint32_t code;
    } stack[2 * HZR_NSYM];
    int sp = 0;
    size_t bits = 0;
    stack[sp].node = (int16_t)root;
    stack[sp].depth = 0;
    stack[sp].code = 0;
    ++sp;
    while (sp > 0) {
        --sp;
        int nd = stack[sp].node;
        unsigned depth = stack[sp].depth;
        uint32_t code = stack[sp].code;
        if (nodes[nd].sym >= 0) {
            if (sink) {
                sink_put(sink, 1, 1);
                sink_put(sink, (uint32_t)nodes[nd].sym, 9);
            }
            bits += 10;
            cb->code[nodes[nd].sym] = code;
            cb->len[nodes[nd].sym] = (uint8_t)depth;
        } else {
            if (sink) sink_put(sink, 0, 1);
            bits += 1;
            /* push b first so a is visited first */
            stack[sp].node = nodes[nd].b;
            stack[sp].depth = (uint8_t)(depth + 1);
            stack[sp].code = code | (1u << depth);
            ++sp;
            stack[sp].node = nodes[nd].a;
            stack[sp].depth = (uint8_t)(depth + 1);
            stack[sp].code = code;
            ++sp;
        }
    }
    return bits;
}

/* Does the block consist of one distinct byte value? (hzr_encode.c:285-305:
 * all zero-type symbols count as one code.) */
static int single_code(const uint32_t hist[HZR_NSYM]) {
    int zero_kind = 0, nonzero = 0;
    for (int s = 0; s < HZR_NSYM; ++s) {
        if (!hist[s]) continue;
        if (s == 0 || s >= 256)
            zero_kind = 1;
        else
            ++nonzero;
    }
    return (zero_kind + nonzero) == 1;
}

static void put_le16(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}
static void put_le32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}
static uint32_t get_le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static uint32_t get_le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static size_t payload_bits(const uint32_t hist[HZR_NSYM], const codebook* cb, size_t tree_bits) {
    size_t bits = tree_bits;
    for (int s = 0; s < HZR_NSYM; ++s) {
        unsigned extra = (s > 256) ? kRunExtra[s - 256] : 0;
        bits += (size_t)hist[s] * (cb->len[s] + extra);
    }
    return bits;
}

void orc_hzr_block_stats(const uint8_t* in, size_t n, uint32_t hist[261], int* mode, size_t* payload_len) {
    token* toks = (token*)malloc(sizeof(token) * (n ? n : 1));
    size_t nt = tokenize(in, n, toks);
    memset(hist, 0, sizeof(uint32_t) * HZR_NSYM);
    for (size_t i = 0; i < nt; ++i) hist[toks[i].sym]++;
    free(toks);
    if (single_code(hist)) {
        *mode = HZR_MODE_FILL;
        *payload_len = 1;
        return;
    }
    codebook cb;
    size_t tb = build_tree(hist, &cb, NULL);
    size_t bytes = (payload_bits(hist, &cb, tb) + 7) / 8;
    if (bytes <= n && bytes < HZR_BLOCK) {
        *mode = HZR_MODE_HUFF;
        *payload_len = bytes;
    } else {
        *mode = HZR_MODE_COPY;
        *payload_len = n;
    }
}

/* One block (hzr_encode.c:369-487).  `room` = bytes left in the output.
 * Returns bytes written (header included) or 0 on failure. */
static size_t encode_block(const uint8_t* in, size_t n, uint8_t* out, size_t room, token* toks) {
    if (room < 7) return 0; /* :384-388 */
    uint32_t hist[HZR_NSYM];
    memset(hist, 0, sizeof(hist));
    size_t nt = tokenize(in, n, toks);
    for (size_t i = 0; i < nt; ++i) hist[toks[i].sym]++;

    if (single_code(hist)) { /* :397-401 -> EncodeFill :341-367 */
        if (room < 8) return 0;
        put_le16(out, 0);
        put_le32(out + 2, orc_crc32c(in, 1));
        out[6] = HZR_MODE_FILL;
        out[7] = in[0];
        return 8;
    }

    codebook cb;
    size_t tree_bits = build_tree(hist, &cb, NULL);
    size_t nbytes = (payload_bits(hist, &cb, tree_bits) + 7) / 8;
    /* the block stream ends at header+in_size or at the output end (:377-382) */
    size_t limit = n;
    if (room - 7 < limit) limit = room - 7;

    if (nbytes <= limit && nbytes < HZR_BLOCK) { /* :403-469 */
        uint8_t* payload = out + 7;
        memset(payload, 0, nbytes);
        bitsink sink = {payload, 0};
        build_tree(hist, &cb, &sink);
        for (size_t i = 0; i < nt; ++i) {
            unsigned s = toks[i].sym;
            sink_put(&sink, cb.code[s], cb.len[s]);
            if (s > 256) sink_put(&sink, toks[i].extra, kRunExtra[s - 256]);
        }
        put_le16(out, (uint32_t)(nbytes - 1)); /* :479-481 */
        put_le32(out + 2, orc_crc32c(payload, nbytes));
        out[6] = HZR_MODE_HUFF;
        return nbytes + 7;
    }

    if (room < n + 7) return 0; /* PlainCopy :307-339 */
    put_le16(out, (uint32_t)(n - 1));
    put_le32(out + 2, orc_crc32c(in, n));
    out[6] = HZR_MODE_COPY;
    memcpy(out + 7, in, n);
    return n + 7;
}

size_t orc_hzr_max_compressed_size(size_t n) {
    size_t blocks = (n + HZR_BLOCK - 1) / HZR_BLOCK;
    return 4 + (n ? n + 7 * blocks : 0);
}

int orc_hzr_encode(const uint8_t* in, size_t n, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!in || !out || !out_len || out_cap < 4) return 0;
    token* toks = (token*)malloc(sizeof(token) * HZR_BLOCK);
    if (!toks) return 0;
    put_le32(out, (uint32_t)n); /* master header :521-522 */
    size_t pos = 4;
    for (size_t off = 0; off < n; off += HZR_BLOCK) {
        size_t bn = n - off < HZR_BLOCK ? n - off : HZR_BLOCK;
        size_t w = encode_block(in + off, bn, out + pos, out_cap - pos, toks);
        if (!w) {
            free(toks);
            return 0;
        }
        pos += w;
    }
    free(toks);
    *out_len = pos;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* decoder (hzr_decode.c:263-333 RecoverTree, :335-567 DecodeSingleBlock)    */
/* ------------------------------------------------------------------------ */
typedef struct {
    const uint8_t* p;
    size_t nbits; /* total bits available */
    size_t pos;
    int bad;
} bitsrc;

static uint32_t src_get(bitsrc* s, unsigned width) {
    if (s->pos + width > s->nbits) {
        s->bad = 1;
        return 0;
    }
    uint32_t v = 0;
    for (unsigned i = 0; i < width; ++i) {
        size_t q = s->pos + i;
        v |= (uint32_t)((s->p[q >> 3] >> (q & 7u)) & 1u) << i;
    }
    s->pos += width;
    return v;
}

typedef struct {
    int16_t a, b;
    int16_t sym;
} dnode;

static int recover_tree(bitsrc* s, dnode* nodes, int* count) {
    /* iterative pre-order rebuild */
    int stack[2 * HZR_NSYM];
    int sp = 0;
    int root = (*count)++;
    stack[sp++] = root;
    /* each stack entry is a node whose description is still to be read; a
     * branch pushes b then a so that a is read first */
    while (sp > 0) {
        int nd = stack[--sp];
        uint32_t leaf = src_get(s, 1);
        if (s->bad) return -1;
        if (leaf) {
            nodes[nd].sym = (int16_t)src_get(s, 9);
            nodes[nd].a = nodes[nd].b = -1;
            if (s->bad) return -1;
        } else {
            if (*count + 2 > 2 * HZR_NSYM - 1) return -1;
            nodes[nd].sym = -1;
            nodes[nd].a = (int16_t)(*count)++;
            nodes[nd].b = (int16_t)(*count)++;
            stack[sp++] = nodes[nd].b;
            stack[sp++] = nodes[nd].a;
        }
    }
    return root;
}

/* returns encoded bytes consumed (header included) or 0 on failure */
static size_t decode_block(const uint8_t* in, size_t avail, uint8_t* out, size_t n) {
    if (avail < 7) return 0;
    size_t enc = get_le16(in) + 1u;
    unsigned mode = in[6];
    if (mode == HZR_MODE_COPY) {
        if (enc != n || avail < 7 + n) return 0;
        memcpy(out, in + 7, n);
        return 7 + n;
    }
    if (mode == HZR_MODE_FILL) {
        if (avail < 8) return 0;
        memset(out, in[7], n);
        return 8;
    }
    if (mode != HZR_MODE_HUFF || avail < 7 + enc) return 0;
    bitsrc s = {in + 7, enc * 8, 0, 0};
    dnode nodes[2 * HZR_NSYM];
    int count = 0;
    int root = recover_tree(&s, nodes, &count);
    if (root < 0) return 0;
    size_t o = 0;
    while (o < n) {
        int nd = root;
        if (nodes[nd].sym >= 0) { /* single-leaf tree: 1-bit codes */
            (void)src_get(&s, 1);
        }
        while (nodes[nd].sym < 0) nd = src_get(&s, 1) ? nodes[nd].b : nodes[nd].a;
        if (s.bad) return 0;
        int sym = nodes[nd].sym;
        if (sym <= 255) {
            out[o++] = (uint8_t)sym;
        } else {
            if (sym > 260) return 0;
            unsigned c = (unsigned)sym - 256;
            size_t z = kRunBase[c] + src_get(&s, kRunExtra[c]);
            if (s.bad || o + z > n) return 0;
            memset(out + o, 0, z);
            o += z;
        }
    }
    /* the reference commits the stream at the byte after the last bit read */
    return 7 + (s.pos + 7) / 8;
}

int orc_hzr_decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* consumed) {
    if (!in || !out || in_len < 4) return 0;
    size_t n = get_le32(in);
    if (out_cap < n) return 0;
    size_t pos = 4;
    for (size_t off = 0; off < n; off += HZR_BLOCK) {
        size_t bn = n - off < HZR_BLOCK ? n - off : HZR_BLOCK;
        size_t used = decode_block(in + pos, in_len - pos, out + off, bn);
        if (!used) return 0;
        pos += used;
    }
    if (consumed) *consumed = pos;
    return 1;
}

int orc_hzr_verify(const uint8_t* in, size_t in_len, size_t* decoded_size) {
    if (!in || in_len < 4) return 0;
    size_t n = get_le32(in);
    if (decoded_size) *decoded_size = n;
    size_t pos = 4;
    for (size_t off = 0; off < n; off += HZR_BLOCK) {
        if (in_len - pos < 7) return 0;
        size_t enc = get_le16(in + pos) + 1u;
        uint32_t crc = get_le32(in + pos + 2);
        if (in[pos + 6] > HZR_MODE_FILL) return 0;
        if (in_len - pos - 7 < enc) return 0;
        if (orc_crc32c(in + pos + 7, enc) != crc) return 0;
        pos += 7 + enc;
    }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* lib_signalpacker/utils.cpp                                               */
/* ------------------------------------------------------------------------ */
void orc_native_to_i32(int32_t* planar, const uint8_t* native, size_t ns, size_t nch, size_t bps) {
    /* utils.cpp:139-141,155-158,171-174,186-189: little-endian, sign-extended
     * from bps bytes; planar[c*ns+s] <- native[(s*nch+c)*bps ..] */
    for (size_t s = 0; s < ns; ++s)
        for (size_t c = 0; c < nch; ++c) {
            const uint8_t* q = native + (s * nch + c) * bps;
            uint32_t u = 0;
            for (size_t k = 0; k < bps; ++k) u |= (uint32_t)q[k] << (8 * k);
            if (bps < 4) {
                unsigned sh = (unsigned)(32 - 8 * bps);
                u <<= sh;
                planar[c * ns + s] = (int32_t)u >> sh;
            } else {
                planar[c * ns + s] = (int32_t)u;
            }
        }
}

void orc_i32_to_native(uint8_t* native, const int32_t* planar, size_t ns, size_t nch, size_t bps) {
    /* utils.cpp:65-73,86-93,105-111,119-120 */
    for (size_t s = 0; s < ns; ++s)
        for (size_t c = 0; c < nch; ++c) {
            uint32_t u = (uint32_t)planar[c * ns + s];
            uint8_t* q = native + (s * nch + c) * bps;
            for (size_t k = 0; k < bps; ++k) q[k] = (uint8_t)(u >> (8 * k));
        }
}

void orc_xdelta_forward(int32_t* a, size_t n) {
    /* delta_encode (utils.cpp:193-202), offset_32(-128) (:215-219),
     * xor_encode_32 (:221-230), each over the flat array; fused here. */
    uint32_t prev_p = 0, prev_o = 0;
    for (size_t i = 0; i < n; ++i) {
        uint32_t p = (uint32_t)a[i];
        uint32_t o = p - prev_p - 128u;
        a[i] = (int32_t)(o ^ prev_o);
        prev_p = p;
        prev_o = o;
    }
}

void orc_xdelta_inverse(int32_t* a, size_t n) {
    /* xor_decode_32 (:232-236), offset_32(+128), delta_decode (:204-213) */
    uint32_t o = 0, p = 0;
    for (size_t i = 0; i < n; ++i) {
        o ^= (uint32_t)a[i];
        p += o + 128u;
        a[i] = (int32_t)p;
    }
}

int32_t orc_average_32(const int32_t* a, size_t len) {
    /* utils.cpp:30-40: `int64 /= size_t` converts the sum to unsigned. */
    int64_t sum = 0;
    for (size_t i = 0; i < len; ++i) sum += a[i];
    uint64_t q = (uint64_t)sum / (uint64_t)len;
    return (int32_t)(int64_t)q;
}

unsigned orc_xdelta_needed_nb(const int32_t* v, size_t n, size_t bps, unsigned nb_min) {
    /* SURVEY.md 8 note a-3: nb passes iff for every v bits [8nb, 8*min(bps,4))
     * equal bit 8nb-1 (sign extension from nb bytes is lossless below 8*bps). */
    unsigned top = (unsigned)(bps < 4 ? bps : 4) * 8;
    unsigned nb = nb_min < 1 ? 1 : nb_min;
    for (; nb < 4; ++nb) {
        unsigned lo = 8 * nb;
        if (lo >= top) break;
        uint32_t mask = (top == 32 ? 0xFFFFFFFFu : ((1u << top) - 1u)) & ~((1u << lo) - 1u);
        int ok = 1;
        for (size_t i = 0; i < n && ok; ++i) {
            uint32_t x = (uint32_t)v[i];
            uint32_t want = ((x >> (lo - 1)) & 1u) ? mask : 0u;
            if ((x & mask) != want) ok = 0;
        }
        if (ok) break;
    }
    return nb;
}

void orc_fwht(int32_t* a, size_t n) {
    /* fwht.c:15-25: stage width from n/2 down to 1; (lo,hi) -> (lo+hi, lo-hi) */
    for (size_t w = n >> 1; w > 0; w >>= 1)
        for (size_t base = 0; base < n; base += 2 * w)
            for (size_t j = base; j < base + w; ++j) {
                uint32_t lo = (uint32_t)a[j], hi = (uint32_t)a[j + w];
                a[j] = (int32_t)(lo + hi);
                a[j + w] = (int32_t)(lo - hi);
            }
}

/* ------------------------------------------------------------------------ */
/* packers                                                                  */
/* ------------------------------------------------------------------------ */
struct orc_packer {
    int kind;
    size_t bps, nch, ns, n;
    unsigned nb;
    int fast_verify;
    int32_t* enc;    /* [nch][ns] (signal_packer_base.h:20) */
    int32_t* tmp;    /* transform scratch */
    uint8_t* planes; /* [4][n]    (signal_packer_base.h:21) */
    uint8_t* verify; /* round-trip buffer */
    float* cos_tab;  /* dct only: [n][n] (signal_packer_dct.cpp:60-74) */
};

static int is_pow2(size_t x) { return x && !(x & (x - 1)); }

orc_packer* orc_packer_new(int kind, size_t bps, size_t nch, size_t ns, size_t nb) {
    if (bps < 1 || bps > 4 || nch == 0 || ns == 0) return NULL;
    if (kind < 0 || kind > 3) return NULL;
    if (nch * ns >= ((size_t)1 << 31)) return NULL;
    orc_packer* p = (orc_packer*)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->kind = kind;
    p->bps = bps;
    p->nch = nch;
    p->ns = ns;
    p->n = nch * ns;
    switch (kind) {
        case ORC_KIND_HZR: p->nb = 4; break;      /* signal_packer_hzr.cpp:39 */
        case ORC_KIND_DCT: p->nb = 2; break;      /* signal_packer_dct.cpp:46 */
        case ORC_KIND_HADAMARD: p->nb = 3; break; /* signal_packer_hadamard.cpp:44 */
        default:
            if (nb < 1 || nb > 4) {
                free(p);
                return NULL;
            }
            p->nb = (unsigned)nb;
    }
    if (kind == ORC_KIND_HADAMARD && !is_pow2(ns)) { /* fwht.c needs n = 2^k */
        free(p);
        return NULL;
    }
    /* dct with ns = 2^k > 8192 (and anything past 32768, where the reference's int table index overflows):
     * the handle is created without the n*n float table, and only the *_coeffs entry points work on it
     * (framing around a transform the caller evaluates in fp64).  Other sizes up to 32768 get the table,
     * as the GPU build gives them: 400 MB at ns = 10000. */
    const int dct_table = kind == ORC_KIND_DCT && (ns <= 8192 || (ns <= 32768 && !is_pow2(ns)));
    p->enc = (int32_t*)calloc(p->n, sizeof(int32_t));
    p->tmp = (int32_t*)calloc(p->ns, sizeof(int32_t));
    p->planes = (uint8_t*)calloc(4 * p->n, 1);
    p->verify = (uint8_t*)calloc(p->n * bps, 1);
    if (dct_table) {
        /* signal_packer_dct.cpp:60-74: COS[x][i] = (float)cos(((2x+1)*i) * PI/(2n)) */
        const double PI = 3.14159265358979323846;
        double step = PI / ((double)ns * 2.0);
        p->cos_tab = (float*)malloc(sizeof(float) * ns * ns);
        if (p->cos_tab)
            for (size_t x = 0; x < ns; ++x)
                for (size_t i = 0; i < ns; ++i) {
                    int arg = (int)((x << 1) * i + i);
                    p->cos_tab[x * ns + i] = (float)cos(arg * step);
                }
    }
    if (!p->enc || !p->tmp || !p->planes || !p->verify || (dct_table && !p->cos_tab)) {
        orc_packer_free(p);
        return NULL;
    }
    return p;
}

void orc_packer_free(orc_packer* p) {
    if (!p) return;
    free(p->enc);
    free(p->tmp);
    free(p->planes);
    free(p->verify);
    free(p->cos_tab);
    free(p);
}

unsigned orc_packer_nb(const orc_packer* p) { return p->nb; }
void orc_packer_set_fast_verify(orc_packer* p, int on) { p->fast_verify = on; }
const int32_t* orc_packer_last_enc(const orc_packer* p) { return p->enc; }

static size_t header_len(const orc_packer* p) {
    return (p->kind == ORC_KIND_DCT || p->kind == ORC_KIND_HADAMARD) ? 3 * p->nch : 0;
}

size_t orc_packer_max_compressed_size(const orc_packer* p) {
    return 1 + header_len(p) + (size_t)p->nb * (4 + orc_hzr_max_compressed_size(p->n));
}

/* signal_packer_base.cpp:38-96 */
static int compress_i32(orc_packer* p, uint8_t* dst, size_t dst_max_len, size_t* dst_len, uint8_t method, unsigned nb,
                        const uint8_t* header, size_t hlen) {
    const size_t n = p->n;
    for (size_t i = 0; i < n; ++i) { /* byte-plane split :40-68 */
        uint32_t v = (uint32_t)p->enc[i];
        for (unsigned k = 0; k < nb; ++k) p->planes[(size_t)k * n + i] = (uint8_t)(v >> (8 * k));
    }
    if (dst_max_len < 1 + hlen) return -1;
    size_t pos = 0;
    dst[pos++] = method; /* :83 */
    if (header && hlen) {
        memcpy(dst + pos, header, hlen); /* :86-91 */
        pos += hlen;
    }
    size_t room = dst_max_len - 1; /* :92 (header bytes are not subtracted) */
    for (unsigned k = 0; k < nb; ++k) { /* :69-82,94-95 */
        if (pos + 4 > dst_max_len) return -1;
        size_t cap = dst_max_len - pos - 4;
        if (room < cap) cap = room; /* hzr_encode is handed `room` as out_size */
        size_t len = 0;
        if (!orc_hzr_encode(p->planes + (size_t)k * n, n, dst + pos + 4, cap, &len)) return -1;
        put_le32(dst + pos, (uint32_t)len);
        pos += 4 + len;
        room -= 4 + len;
    }
    *dst_len = pos;
    return 0;
}

/* signal_packer_base.cpp:98-139 */
static int decompress_i32(orc_packer* p, const uint8_t* src, size_t* src_len, uint8_t* method, unsigned nb, uint8_t* header,
                          size_t hlen) {
    const size_t n = p->n;
    size_t pos = 0;
    *method = src[pos++];
    if (header && hlen) {
        memcpy(header, src + pos, hlen);
        pos += hlen;
    }
    memset(p->planes, 0, 4 * n); /* serialized_.fill(0) :117 */
    for (unsigned k = 0; k < nb; ++k) {
        size_t len = get_le32(src + pos);
        pos += 4;
        (void)orc_hzr_decode(src + pos, len, p->planes + (size_t)k * n, n, NULL); /* status ignored :106 */
        pos += len;
    }
    *src_len = pos;
    unsigned sh = 32 - 8 * nb;
    for (size_t i = 0; i < n; ++i) { /* :121-138: sign-extend from nb bytes */
        uint32_t v = 0;
        for (unsigned k = 0; k < nb; ++k) v |= (uint32_t)p->planes[(size_t)k * n + i] << (8 * k);
        p->enc[i] = nb < 4 ? ((int32_t)(v << sh) >> sh) : (int32_t)v;
    }
    return 0;
}

static void means_header(const int32_t* means, size_t nch, uint8_t* h) {
    for (size_t c = 0; c < nch; ++c) { /* hadamard.cpp:73-79, dct.cpp:120-126 */
        h[3 * c + 0] = (uint8_t)means[c];
        h[3 * c + 1] = (uint8_t)((uint32_t)means[c] >> 8);
        h[3 * c + 2] = (uint8_t)((uint32_t)means[c] >> 16);
    }
}

static int32_t mean_from_header(const uint8_t* h, size_t c) {
    uint32_t u = (uint32_t)h[3 * c] | ((uint32_t)h[3 * c + 1] << 8) | ((uint32_t)h[3 * c + 2] << 16);
    return (int32_t)(u << 8) >> 8;
}

static void remove_means(orc_packer* p, int32_t* means) {
    for (size_t c = 0; c < p->nch; ++c) {
        int32_t* row = p->enc + c * p->ns;
        means[c] = orc_average_32(row, p->ns);
        for (size_t s = 0; s < p->ns; ++s) row[s] = (int32_t)((uint32_t)row[s] - (uint32_t)means[c]);
    }
}

static void add_means(orc_packer* p, const uint8_t* header) {
    for (size_t c = 0; c < p->nch; ++c) {
        int32_t m = mean_from_header(header, c);
        int32_t* row = p->enc + c * p->ns;
        for (size_t s = 0; s < p->ns; ++s) row[s] = (int32_t)((uint32_t)row[s] + (uint32_t)m);
    }
}

/* signal_packer_dct.cpp:76-87.  `int * float` multiplies in float, the sum
 * runs in double, the scale is (double)Cs*sqrt(2/n)/128, the store truncates. */
static void dct_forward(const orc_packer* p, const int32_t* src, int32_t* dst) {
    const size_t n = p->ns;
    const double ratio1 = sqrt(2.0 / (double)n);
    const float cs0 = (float)(1 / sqrt(2));
    for (size_t i = 0; i < n; ++i) {
        double sum = 0;
        for (size_t x = 0; x < n; ++x) {
            float prod = (float)src[x] * p->cos_tab[x * n + i];
            sum += prod;
        }
        float cs = i ? 1.0f : cs0;
        sum *= cs * ratio1 / 128.0;
        dst[i] = (int32_t)sum;
    }
}

/* signal_packer_dct.cpp:89-100 */
static void dct_inverse(const orc_packer* p, const int32_t* src, int32_t* dst) {
    const size_t n = p->ns;
    const double ratio1 = sqrt(2.0 / (double)n);
    const float cs0 = (float)(1 / sqrt(2));
    for (size_t i = 0; i < n; ++i) {
        double sum = 0;
        for (size_t x = 0; x < n; ++x) {
            float cs = x ? 1.0f : cs0;
            float prod = cs * (float)src[x] * p->cos_tab[i * n + x];
            sum += prod;
        }
        sum *= ratio1 * 128.0;
        dst[i] = (int32_t)sum;
    }
}

/* Framing of the lossy packers around externally evaluated transform coefficients: what
 * signal_packer_dct.cpp:117-127 / signal_packer_hadamard.cpp:73-80 do after the transform.
 * coeffs = [nch][ns] transform output (dct: before the flat delta/xor), means = [nch]. */
int orc_packer_compress_coeffs(orc_packer* p, const int32_t* coeffs, const int32_t* means, uint8_t* dst, size_t dst_max_len,
                               size_t* dst_len) {
    if (p->kind != ORC_KIND_DCT && p->kind != ORC_KIND_HADAMARD) return -3;
    size_t hlen = header_len(p);
    uint8_t* header = (uint8_t*)malloc(hlen);
    memcpy(p->enc, coeffs, sizeof(int32_t) * p->n);
    if (p->kind == ORC_KIND_DCT) orc_xdelta_forward(p->enc, p->n); /* dct.cpp:117-119 */
    means_header(means, p->nch, header);
    int rc = compress_i32(p, dst, dst_max_len, dst_len, p->kind == ORC_KIND_DCT ? 1 : 2, p->nb, header, hlen);
    free(header);
    return rc;
}

/* Inverse of the above: dct.cpp:130-139 / hadamard.cpp:83-92 up to (not including) the inverse transform. */
int orc_packer_decompress_coeffs(orc_packer* p, const uint8_t* src, size_t* src_len, int32_t* coeffs, int32_t* means) {
    if (p->kind != ORC_KIND_DCT && p->kind != ORC_KIND_HADAMARD) return -3;
    uint8_t method = 0;
    size_t hlen = header_len(p);
    uint8_t* header = (uint8_t*)malloc(hlen);
    decompress_i32(p, src, src_len, &method, p->nb, header, hlen);
    if (p->kind == ORC_KIND_DCT) orc_xdelta_inverse(p->enc, p->n);
    memcpy(coeffs, p->enc, sizeof(int32_t) * p->n);
    for (size_t c = 0; c < p->nch; ++c) means[c] = mean_from_header(header, c);
    free(header);
    return 0;
}

int orc_packer_decompress(orc_packer* p, const uint8_t* src, size_t* src_len, uint8_t* dst) {
    if (p->kind == ORC_KIND_DCT && !p->cos_tab) return -3;
    uint8_t method = 0;
    size_t hlen = header_len(p);
    uint8_t* header = hlen ? (uint8_t*)malloc(hlen) : NULL;
    decompress_i32(p, src, src_len, &method, p->nb, header, hlen);
    switch (p->kind) {
        case ORC_KIND_HZR: /* signal_packer_hzr.cpp:57-65 */
            break;
        case ORC_KIND_XDELTA_HZR: /* signal_packer_xdelta_hzr.cpp:74-85 */
            orc_xdelta_inverse(p->enc, p->n);
            break;
        case ORC_KIND_HADAMARD: /* signal_packer_hadamard.cpp:83-104 */
            for (size_t c = 0; c < p->nch; ++c) orc_fwht(p->enc + c * p->ns, p->ns); /* normalize2(ratio 1) is a no-op */
            add_means(p, header);
            break;
        case ORC_KIND_DCT: /* signal_packer_dct.cpp:130-153 */
            orc_xdelta_inverse(p->enc, p->n);
            for (size_t c = 0; c < p->nch; ++c) {
                dct_inverse(p, p->enc + c * p->ns, p->tmp);
                memcpy(p->enc + c * p->ns, p->tmp, sizeof(int32_t) * p->ns);
            }
            add_means(p, header);
            break;
    }
    orc_i32_to_native(dst, p->enc, p->ns, p->nch, p->bps);
    free(header);
    return 0;
}

int orc_packer_compress(orc_packer* p, const uint8_t* src, uint8_t* dst, size_t dst_max_len, size_t* dst_len) {
    if (p->kind == ORC_KIND_DCT && !p->cos_tab) return -3;
    size_t hlen = header_len(p);
    int rc = 0;
    switch (p->kind) {
        case ORC_KIND_HZR: /* signal_packer_hzr.cpp:51-55 */
            orc_native_to_i32(p->enc, src, p->ns, p->nch, p->bps);
            return compress_i32(p, dst, dst_max_len, dst_len, 0, p->nb, NULL, 0);

        case ORC_KIND_XDELTA_HZR: /* signal_packer_xdelta_hzr.cpp:52-72 */
            for (;;) {
                orc_native_to_i32(p->enc, src, p->ns, p->nch, p->bps);
                orc_xdelta_forward(p->enc, p->n);
                if (p->fast_verify) {
                    unsigned need = orc_xdelta_needed_nb(p->enc, p->n, p->bps, p->nb);
                    p->nb = need;
                    return compress_i32(p, dst, dst_max_len, dst_len, 0, p->nb, NULL, 0);
                }
                rc = compress_i32(p, dst, dst_max_len, dst_len, 0, p->nb, NULL, 0);
                if (rc) return rc;
                size_t used = 0; /* round-trip self check :59-62 */
                orc_packer_decompress(p, dst, &used, p->verify);
                if (memcmp(src, p->verify, p->n * p->bps) == 0) return 0;
                if (p->nb >= 4) return -2; /* cannot happen: nb=4 is lossless */
                p->nb++;                   /* :63-69 */
            }

        case ORC_KIND_HADAMARD: { /* signal_packer_hadamard.cpp:57-81 */
            int32_t* means = (int32_t*)malloc(sizeof(int32_t) * p->nch);
            uint8_t* header = (uint8_t*)malloc(hlen);
            orc_native_to_i32(p->enc, src, p->ns, p->nch, p->bps);
            remove_means(p, means);
            for (size_t c = 0; c < p->nch; ++c) {
                int32_t* row = p->enc + c * p->ns;
                orc_fwht(row, p->ns);
                /* fwht_normalize (fwht.c:30-34): int /= (n / 1.0), i.e. the
                 * quotient is formed in double and truncated toward zero */
                for (size_t s = 0; s < p->ns; ++s) row[s] = (int32_t)((double)row[s] / ((double)(int)p->ns / 1.0));
            }
            means_header(means, p->nch, header);
            rc = compress_i32(p, dst, dst_max_len, dst_len, 2, p->nb, header, hlen);
            free(means);
            free(header);
            return rc;
        }

        case ORC_KIND_DCT: { /* signal_packer_dct.cpp:102-128 */
            int32_t* means = (int32_t*)malloc(sizeof(int32_t) * p->nch);
            uint8_t* header = (uint8_t*)malloc(hlen);
            orc_native_to_i32(p->enc, src, p->ns, p->nch, p->bps);
            remove_means(p, means);
            for (size_t c = 0; c < p->nch; ++c) {
                dct_forward(p, p->enc + c * p->ns, p->tmp);
                memcpy(p->enc + c * p->ns, p->tmp, sizeof(int32_t) * p->ns);
            }
            orc_xdelta_forward(p->enc, p->n); /* :117-119 */
            means_header(means, p->nch, header);
            rc = compress_i32(p, dst, dst_max_len, dst_len, 1, p->nb, header, hlen);
            free(means);
            free(header);
            return rc;
        }
    }
    return -1;
}

double orc_prdn(const uint8_t* orig_native, const uint8_t* dec_native, size_t ns, size_t nch, size_t bps) {
    /* rspt_test.cpp:98-111 */
    int32_t* o = (int32_t*)malloc(sizeof(int32_t) * ns * nch);
    int32_t* d = (int32_t*)malloc(sizeof(int32_t) * ns * nch);
    orc_native_to_i32(o, orig_native, ns, nch, bps);
    orc_native_to_i32(d, dec_native, ns, nch, bps);
    double mse = 0, ref = 0;
    for (size_t c = 0; c < nch; ++c) {
        int32_t mean = orc_average_32(o + c * ns, ns);
        for (size_t s = 0; s < ns; ++s) {
            double t = (double)(int32_t)((uint32_t)o[c * ns + s] - (uint32_t)d[c * ns + s]);
            mse += t * t;
            uint32_t dm = (uint32_t)o[c * ns + s] - (uint32_t)mean;
            ref += (double)(int32_t)(dm * dm); /* int*int in the reference */
        }
    }
    free(o);
    free(d);
    return sqrt(mse / ref) * 100.0;
}


/* ===========================================================================
 * IIR pre-filter: the step in front of the packers in the reference's own pipeline
 * (lib_rspt_test/rspt_test.cpp:116-136), lib_rspt/lib_filter/iir_filter.cpp:46-116.
 * Double arithmetic in exactly the reference's order of operations (build with
 * -ffp-contract=off: no fused multiply-add, as in the reference's x86-64 build).
 * ===========================================================================*/
typedef struct {
    double x[5], y[5], n[5], d[5];
    size_t nc;
} orc_iir;

static void iir_shift(orc_iir* f, double x) { /* iir_filter.cpp:66-71, 81-86 */
    for (size_t i = f->nc - 1; i > 0; --i) {
        f->x[i] = f->x[i - 1];
        f->y[i] = f->y[i - 1];
    }
    f->x[0] = x;
}

static double iir_filter(orc_iir* f, double x) { /* i_filter::filter, iir_filter.cpp:64-77: terms added one by one, feed-forward and feedback interleaved */
    iir_shift(f, x);
    double acc = f->d[0] * f->x[0];
    for (size_t i = 1; i < f->nc; ++i) {
        acc += f->d[i] * f->x[i];
        acc -= f->n[i] * f->y[i];
    }
    f->y[0] = acc;
    return acc;
}

static double iir_filter_opt(orc_iir* f, double x) { /* i_filter::filter_opt, iir_filter.cpp:79-104 with :23-41: one expression, left to right */
    iir_shift(f, x);
    const double *d = f->d, *n = f->n, *xz = f->x;
    double* yz = f->y;
    switch (f->nc) {
        case 5: yz[0] = d[0] * xz[0] + d[1] * xz[1] + d[2] * xz[2] + d[3] * xz[3] + d[4] * xz[4] - n[1] * yz[1] - n[2] * yz[2] - n[3] * yz[3] - n[4] * yz[4]; break;
        case 4: yz[0] = d[0] * xz[0] + d[1] * xz[1] + d[2] * xz[2] + d[3] * xz[3] - n[1] * yz[1] - n[2] * yz[2] - n[3] * yz[3]; break;
        case 3: yz[0] = d[0] * xz[0] + d[1] * xz[1] + d[2] * xz[2] - n[1] * yz[1] - n[2] * yz[2]; break;
        case 2: yz[0] = d[0] * xz[0] + d[1] * xz[1] - n[1] * yz[1]; break;
        default: break; /* (the reference leaves y[0] as shifted: not a filter; callers pass 2..5) */
    }
    return yz[0];
}

int orc_iir_prefilter_native(uint8_t* native, size_t bps, size_t nch, size_t ns, const double* n, const double* d, size_t nc, int init_nr_samples,
                             int shared_state) {
    /* rspt_test.cpp:118-135: native -> [nch][ns] int32, ONE filter object for all channels (its state runs on from
     * channel to channel: init_history_values feeds 4*nr_samples copies of the channel's first sample through
     * filter(), which damps the old state but does not erase it), filter_opt per sample, result truncated to int32,
     * back to native.  shared_state = 0: a fresh filter per channel (what the GPU's channel-parallel mode computes). */
    if (nc < 2 || nc > 5 || bps < 1 || bps > 4) return -1;
    int32_t* planar = (int32_t*)malloc(sizeof(int32_t) * nch * ns);
    if (!planar) return -2;
    orc_native_to_i32(planar, native, ns, nch, bps);
    orc_iir f;
    memset(&f, 0, sizeof f);
    f.nc = nc;
    memcpy(f.n, n, nc * sizeof(double));
    memcpy(f.d, d, nc * sizeof(double));
    for (size_t c = 0; c < nch; ++c) {
        int32_t* row = planar + c * ns;
        if (!shared_state) {
            memset(f.x, 0, sizeof f.x);
            memset(f.y, 0, sizeof f.y);
        }
        for (int i = 0; i < 4 * init_nr_samples; ++i) iir_filter(&f, (double)row[0]); /* init_history_values :106-110 */
        for (size_t s = 0; s < ns; ++s) row[s] = (int32_t)iir_filter_opt(&f, (double)row[s]);
    }
    orc_i32_to_native(native, planar, ns, nch, bps);
    free(planar);
    return 0;
}
