// The multi-GPU path from C++ through the C ABI alone (include/rspt_hip.h): one host thread, one packer handle and one
// ncclComm_t per visible GPU; every rank compresses its contiguous shard of independent blocks on its device
// (rspt_hip_compress_batch_dev), packs the streams into a container (rspt_hip_pack_batch_dev) and the containers travel to
// rank 0 over RCCL (rspt_hip_gather_containers: ncclAllGather of the sizes, one group of ncclSend / ncclRecv for the payload).
// Rank 0 decodes every gathered container on its own device (rspt_hip_decompress_packed_dev) and compares with the input.
// Then three steps of the lagged form (rspt_hip_gather_post_sizes / _post_payload / _wait: no host synchronisation in a step,
// the payload of step i posted during step i + 1), the containers of the last two steps decoded on rank 0.
// Runs with however many devices are visible (a world of one included).  Exit code 0 = all good.  A rank that fails ends the
// whole process at once: its peers would otherwise wait in a collective it never enters.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "rspt_hip.h"

int main() {
    const int world = rspt_hip_device_count();
    if (world <= 0) {
        std::fprintf(stderr, "no gfx950 device\n");
        return 2;
    }
    const size_t bps = 4, nch = 12, ns = 8192, nblocks = 24, block_bytes = bps * nch * ns;
    std::vector<int32_t> input(nblocks * nch * ns);
    for (size_t b = 0; b < nblocks; ++b)
        for (size_t s = 0; s < ns; ++s)
            for (size_t c = 0; c < nch; ++c)
                input[(b * ns + s) * nch + c] = (int32_t)(((s * (c + 1) + 31 * b) % 977) - 400 + ((s ^ b) & 7) + (b == 5 && s == 100 ? 1 << 20 : 0));
    std::vector<ncclComm_t> comms(world);
    std::vector<int> devs(world);
    for (int d = 0; d < world; ++d) devs[d] = d;
    if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) {
        std::fprintf(stderr, "ncclCommInitAll failed\n");
        return 2;
    }
    std::vector<int> fail(world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r]() {
            auto check = [&](bool ok, const char* what) {
                if (!ok) {
                    fail[r] = 1;
                    std::fprintf(stderr, "rank %d: %s failed\n", r, what);
                    std::fflush(stderr);
                    std::_Exit(1);  // (never leave the other ranks waiting in a collective)
                }
                return ok;
            };
            hipSetDevice(r);
            rspt_hip_packer* pk = nullptr;
            if (!check(rspt_hip_packer_create(&pk, RSPT_HIP_KIND_XDELTA_HZR, bps, nch, ns, 2, r) == RSPT_HIP_OK, "create")) return;
            const size_t first = (size_t)r * nblocks / world, count = (size_t)(r + 1) * nblocks / world - first;  // contiguous shard
            hipStream_t st = (hipStream_t)rspt_hip_stream(pk);
            const size_t stride = (rspt_hip_max_compressed_size(pk) + 255) / 256 * 256, bound = rspt_hip_pack_bound(pk, nblocks);
            uint8_t *d_src, *d_dst, *d_packed, *d_recv = nullptr;
            uint64_t *d_sizes, *d_total, *h_totals;
            hipMalloc(&d_src, count * block_bytes + 64);
            hipMalloc(&d_dst, count * stride);
            hipMalloc(&d_packed, bound);
            hipMalloc(&d_sizes, count * 8);
            hipMalloc(&d_total, 8);
            hipHostMalloc((void**)&h_totals, world * 8, hipHostMallocDefault);
            if (r == 0) hipMalloc(&d_recv, (size_t)world * bound);
            hipMemcpyAsync(d_src, input.data() + first * nch * ns, count * block_bytes, hipMemcpyHostToDevice, st);
            check(rspt_hip_compress_batch_dev(pk, d_src, count, d_dst, stride, d_sizes, st) == RSPT_HIP_OK, "compress_batch");
            check(rspt_hip_pack_batch_dev(pk, d_dst, stride, d_sizes, count, d_packed, d_total, st) == RSPT_HIP_OK, "pack_batch");
            check(rspt_hip_gather_containers(pk, comms[r], r, world, 0, d_packed, d_total, d_recv, bound, h_totals, st) == RSPT_HIP_OK, "gather");
            check(hipStreamSynchronize(st) == hipSuccess, "sync");
            if (r == 0 && !fail[r]) {
                // the consumer side: every rank's container decodes on this device, streams in block order
                uint8_t* d_back;
                uint64_t* d_used;
                hipMalloc(&d_back, nblocks * block_bytes);
                hipMalloc(&d_used, nblocks * 8);
                std::vector<uint8_t> back(block_bytes * nblocks);
                for (int q = 0; q < world && !fail[r]; ++q) {
                    const size_t qf = (size_t)q * nblocks / world, qc = (size_t)(q + 1) * nblocks / world - qf;
                    check(h_totals[q] >= 32 + 16 * qc && h_totals[q] <= bound, "container length");
                    check(rspt_hip_decompress_packed_dev(pk, d_recv + (size_t)q * bound, h_totals[q], qc, d_back, d_used, st) == RSPT_HIP_OK, "decompress_packed");
                    hipMemcpyAsync(back.data(), d_back, qc * block_bytes, hipMemcpyDeviceToHost, st);
                    hipStreamSynchronize(st);
                    check(std::memcmp(back.data(), input.data() + qf * nch * ns, qc * block_bytes) == 0, "round trip of a gathered shard");
                }
                hipFree(d_back);
                hipFree(d_used);
            }
            // ---- the lagged form, three steps: two container buffers and (on rank 0) two receive areas alternate ----
            {
                uint8_t *packed2[2] = {d_packed, nullptr}, *recv2[2] = {d_recv, nullptr};
                uint64_t* total2[2] = {d_total, nullptr};
                hipMalloc(&packed2[1], bound);
                hipMalloc(&total2[1], 8);
                if (r == 0) hipMalloc(&recv2[1], (size_t)world * bound);
                std::vector<uint64_t> ht(2 * world);
                const int steps = 3;
                for (int i = 0; i < steps; ++i) {
                    const int slot = i & 1;
                    check(rspt_hip_gather_wait(pk, slot, st) == RSPT_HIP_OK, "gather_wait");  // the payload of step i - 2 has left packed2[slot]
                    check(rspt_hip_compress_batch_dev(pk, d_src, count, d_dst, stride, d_sizes, st) == RSPT_HIP_OK, "compress_batch (lagged)");
                    check(rspt_hip_pack_batch_dev(pk, d_dst, stride, d_sizes, count, packed2[slot], total2[slot], st) == RSPT_HIP_OK, "pack_batch (lagged)");
                    if (i) check(rspt_hip_gather_post_payload(pk, comms[r], r, world, 0, packed2[slot ^ 1], slot ^ 1, recv2[slot ^ 1], bound, &ht[(slot ^ 1) * world]) == RSPT_HIP_OK, "post_payload");
                    check(rspt_hip_gather_post_sizes(pk, comms[r], world, total2[slot], slot, st) == RSPT_HIP_OK, "post_sizes");
                }
                const int last = (steps - 1) & 1;
                check(rspt_hip_gather_post_payload(pk, comms[r], r, world, 0, packed2[last], last, recv2[last], bound, &ht[last * world]) == RSPT_HIP_OK, "post_payload (flush)");
                for (int slot = 0; slot < 2; ++slot) check(rspt_hip_gather_wait(pk, slot, st) == RSPT_HIP_OK, "gather_wait (end)");
                check(hipStreamSynchronize(st) == hipSuccess, "sync (lagged)");
                if (r == 0) {
                    uint8_t* d_back;
                    uint64_t* d_used;
                    hipMalloc(&d_back, nblocks * block_bytes);
                    hipMalloc(&d_used, nblocks * 8);
                    std::vector<uint8_t> back(block_bytes * nblocks);
                    for (int slot = 0; slot < 2; ++slot)
                        for (int q = 0; q < world; ++q) {
                            const size_t qf = (size_t)q * nblocks / world, qc = (size_t)(q + 1) * nblocks / world - qf;
                            check(ht[slot * world + q] == ht[q] && ht[q] >= 32 + 16 * qc && ht[q] <= bound, "sizes of the lagged steps");  // (same input, same carried nb)
                            check(rspt_hip_decompress_packed_dev(pk, recv2[slot] + (size_t)q * bound, ht[slot * world + q], qc, d_back, d_used, st) == RSPT_HIP_OK, "decompress_packed (lagged)");
                            hipMemcpyAsync(back.data(), d_back, qc * block_bytes, hipMemcpyDeviceToHost, st);
                            hipStreamSynchronize(st);
                            check(std::memcmp(back.data(), input.data() + qf * nch * ns, qc * block_bytes) == 0, "round trip of a shard gathered by the lagged form");
                        }
                    hipFree(d_back);
                    hipFree(d_used);
                }
                hipFree(packed2[1]); hipFree(total2[1]); hipFree(recv2[1]);
            }
            hipFree(d_src); hipFree(d_dst); hipFree(d_packed); hipFree(d_sizes); hipFree(d_total); hipFree(d_recv);
            hipHostFree(h_totals);
            rspt_hip_packer_destroy(pk);
        });
    for (auto& t : th) t.join();
    int bad = 0;
    for (int r = 0; r < world; ++r) bad |= fail[r];
    for (int r = 0; r < world; ++r) ncclCommDestroy(comms[r]);
    std::printf("%d rank(s), %zu blocks, %s\n", world, nblocks, bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
