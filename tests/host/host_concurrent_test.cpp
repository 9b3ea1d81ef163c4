// Two host threads issue MatMult4Stream at the same time on ONE shared key set, the way GenoBlockMult runs
// assoc_num_blocks_parallel calls concurrently (assoc.go:360-408, each with its own evaluator: matmult.go:1371).
// Each thread owns a fork of the CryptoParams (own HIP queues / scratch, shared tables + keys).  Every call's output is written
// out and compared with the oracle by tests/test_host_mirror.py; results must not depend on what the other thread is doing.
// Usage: host_concurrent_test <casedir> <nthreads> <rounds>.  Inputs as host_gpu_test plus geno<t>.bin / A<t>.bin per thread.
#include "../../sfgwas_amd/host/gwas.hpp"
#include <atomic>
#include <fstream>
#include <iostream>
#include <thread>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
static void writeU64(const std::string &fn, const std::vector<uint64_t> &v) { std::ofstream f(fn, std::ios::binary); f.write((const char *)v.data(), v.size() * 8); }
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1]; const int nthreads = atoi(argv[2]), rounds = atoi(argv[3]);
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, 17179869184.0);
        const int N = cps->N(), beta = (nq + np - 1) / np; const size_t kw = (size_t)beta * 2 * (nq + np) * N;
        auto keys = readU64(dir + "/keys.bin");
        for (size_t k = 0, off = 1; k < keys[0]; k++, off += 1 + kw) crypto::LoadRotationKey(cps.get(), keys[off], std::vector<uint64_t>(keys.begin() + off + 1, keys.begin() + off + 1 + kw), false);
        std::atomic<int> failures{0};
        std::vector<std::string> errors(nthreads);
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back([&, t]() {
            try {
                std::ifstream cs(dir + "/case" + std::to_string(t) + ".txt"); uint64_t nrow, ncol; int s, level, maxLevel, square; cs >> nrow >> ncol >> s >> level >> maxLevel >> square;
                auto mine = cps->Fork();                                         // private queues + scratch, shared keys
                const int slots = mine->GetSlots(), nbr = (int)((nrow - 1) / slots) + 1;
                crypto::CipherMatrix A = gwas::unflatten(readU64(dir + "/A" + std::to_string(t) + ".bin"), s, nbr, level, 17179869184.0, N);
                std::vector<uint64_t> first;
                for (int r = 0; r < rounds; r++) {
                    gwas::GenoFileStream gfs(dir + "/geno" + std::to_string(t) + ".bin", nrow, ncol, false);
                    auto [out, sum, sq] = gwas::MatMult4Stream(mine.get(), A, &gfs, maxLevel, false, square != 0, 0);
                    auto flat = gwas::flattenCipherMatrix(out);
                    if (r == 0) first = flat;
                    else if (flat != first) throw std::runtime_error("result changed between rounds while another thread was running");
                }
                writeU64(dir + "/out_thr" + std::to_string(t) + ".bin", first);
            } catch (const std::exception &e) { errors[t] = e.what(); failures++; }
        });
        for (auto &x : th) x.join();
        if (failures) { for (auto &e : errors) if (!e.empty()) std::cerr << "ERROR: " << e << std::endl; return 1; }
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
