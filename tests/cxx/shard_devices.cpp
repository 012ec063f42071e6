// One packer per visible GPU, one host thread per packer, a contiguous shard of independent blocks each:
// the C++ side of the multi-GPU path, through nothing but include/signal_packer.h (+ rspt_hip_device_count()).
// Every thread compresses its shard block by block, decompresses it again and checks the round trip; the streams of
// equal blocks must not depend on which device produced them.  Exit code 0 = all good.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "rspt_hip.h"
#include "signal_packer.h"

static uint32_t fnv1a(const unsigned char* p, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 16777619u;
    return h;
}

int main() {
    const int ndev = rspt_hip_device_count();
    if (ndev <= 0) {
        std::fprintf(stderr, "no gfx950 device\n");
        return 2;
    }
    const size_t bps = 4, nch = 12, ns = 8192, nblocks = 16, block_bytes = bps * nch * ns;
    std::vector<std::vector<int32_t>> blocks(nblocks, std::vector<int32_t>(nch * ns));
    for (size_t b = 0; b < nblocks; ++b)  // block b: per-channel ramps with a little structure, the same for b and b + 8
        for (size_t s = 0; s < ns; ++s)
            for (size_t c = 0; c < nch; ++c) blocks[b][s * nch + c] = (int32_t)(((s * (c + 1)) % 977) - 400 + ((s ^ (b % 8)) & 7));
    std::vector<uint32_t> hashes(nblocks, 0);
    std::vector<int> fail(ndev, 0);
    std::vector<std::thread> th;
    for (int d = 0; d < ndev; ++d)
        th.emplace_back([&, d]() {
            rspt_cxx_set_device(d);  // the packers this thread creates live on GPU d
            i_signal_packer* pk = i_signal_packer::new_xdelta_hzr(bps, nch, ns, 3);
            const size_t first = d * nblocks / ndev, last = (d + 1) * nblocks / ndev;  // contiguous shard
            std::vector<unsigned char> dst(2 * block_bytes), back(block_bytes);
            for (size_t b = first; b < last; ++b) {
                size_t len = 0, used = 0;
                pk->compress((const unsigned char*)blocks[b].data(), dst.data(), dst.size(), len);
                if (len == 0) { fail[d] = 1; break; }
                hashes[b] = fnv1a(dst.data(), len);
                pk->decompress(dst.data(), used, back.data());
                if (used != len || std::memcmp(back.data(), blocks[b].data(), block_bytes) != 0) { fail[d] = 1; break; }
            }
            i_signal_packer::delete_xdelta_hzr(pk);
        });
    for (auto& t : th) t.join();
    int bad = 0;
    for (int d = 0; d < ndev; ++d) bad |= fail[d];
    for (size_t b = 0; b + 8 < nblocks; ++b) bad |= hashes[b] != hashes[b + 8];  // equal blocks, (possibly) different devices
    std::printf("%d device(s), %zu blocks, %s\n", ndev, nblocks, bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
