// The multi-GPU host loop of INTEGRATION.md ("Several GPUs in one host process"), as a program: one engine per device, each
// with its share of the streams, one worker thread per device.  Built and run WITHOUT a GPU by tests/test_host_and_abi_cpu.py
// (cpq_engine_create then fails with CPQ_ERR_NO_DEVICE on every device and the program says so: what is checked is that the
// snippet compiles against the C header alone and links against the library's exported symbols).
#include "convopeq_mi355x.h"

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

int main(int argc, char** argv)
{
    const int nGpus = argc > 1 ? std::atoi(argv[1]) : 2;
    const int streamsPerGpu = 4, block = 512, blocksPerCall = 4, n = block * blocksPerCall;
    std::vector<cpq_engine*> eng((size_t)nGpus, nullptr);
    int created = 0;
    for (int g = 0; g < nGpus; ++g) {
        cpq_engine_desc d = {};
        d.struct_size = (int32_t)sizeof d;
        d.device = g;                       // one engine per GPU; global stream s lives on GPU s / streamsPerGpu
        d.n_streams = streamsPerGpu;
        d.block_size = block;
        d.max_ir_len = 4096;
        d.max_blocks_per_call = blocksPerCall;
        d.sample_rate = 48000.0;
        const int32_t rc = cpq_engine_create(&d, &eng[(size_t)g]);
        if (rc != CPQ_OK) { std::printf("device %d: status %d (%s)\n", g, (int)rc, cpq_last_error(nullptr)); continue; }
        ++created;
    }
    // per audio callback: one worker per GPU, each on its own slice of the planar [channel][sample] buffers
    std::vector<double> in((size_t)nGpus * streamsPerGpu * 2 * n, 0.0), out(in.size(), 0.0);
    std::vector<std::thread> workers;
    for (int g = 0; g < nGpus; ++g) {
        if (!eng[(size_t)g]) continue;
        workers.emplace_back([&, g] {
            const size_t off = (size_t)g * streamsPerGpu * 2 * n;
            (void)cpq_engine_process_block(eng[(size_t)g], in.data() + off, out.data() + off, n);
        });
    }
    for (auto& w : workers) w.join();
    for (cpq_engine* e : eng) if (e) cpq_engine_destroy(e);
    std::printf("engines created: %d of %d\n", created, nGpus);
    return 0;
}
