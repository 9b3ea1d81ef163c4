// mpc.CSigmoidApprox's local computation (mhe.go:634-667, called at gwas/assoc.go:1045) on device-resident ciphertexts with REAL keys from a toy secret:
// change of variable (MultByConstNew 2/(b-a), Rescale, AddConst) + eval.EvaluateCheby of ckks.Approximate(Sigmoid, A, B, Degree).
// Dumps the interpolation coefficients, the result words, level and scale; tests/test_host_mirror.py replays every evaluator step with the oracle and
// decrypts the result against the sigmoid.  The oracle is linked ONLY to derive the relinearisation key from the secret.
// Usage: host_sigmoid_test <casedir>
#include "../../sfgwas_amd/host/gwas.hpp"
#include "../../oracle/sfgwas_oracle.h"
#include <fstream>
#include <iostream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::ifstream cs(dir + "/case.txt"); int nct, level, degree; double A, B; uint64_t secretSeed; cs >> nct >> level >> degree >> A >> B >> secretSeed;
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        const double SC = 17179869184.0;
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, SC);
        const int N = cps->N();
        orc_ring *ring = orc_ring_new(14, nq, np, mod.data() + 2, nullptr);
        std::vector<int8_t> sec(N); orc_gen_secret(ring, secretSeed, sec.data());
        std::vector<uint64_t> key((size_t)orc_rotkeys_beta(ring) * 2 * (nq + np) * N);
        orc_gen_rlk(ring, sec.data(), 4999, key.data()); crypto::LoadRelinKey(cps.get(), key, false);
        orc_ring_free(ring);
        crypto::DevCipherVector x = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/x.bin"), 1, nct, level, SC, N)[0]);
        if (crypto::SigmoidNeedsBootstrap(x.level, degree)) throw std::runtime_error("the input level would need a bootstrap first");
        const crypto::ChebyshevInterpolation ch = crypto::Approximate(crypto::Sigmoid, A, B, degree);
        { std::ofstream f(dir + "/coeffs.bin", std::ios::binary); f.write((const char *)ch.poly.c.data(), ch.poly.c.size() * 8); }
        crypto::DevCipherVector y = crypto::CSigmoidApproxLocal(cps.get(), x, A, B, degree);
        { std::ofstream f(dir + "/y.bin", std::ios::binary); auto w = gwas::flattenCipherMatrix({crypto::ToHost(y)}); f.write((const char *)w.data(), w.size() * 8); }
        std::ofstream meta(dir + "/meta.txt"); meta.precision(17); meta << y.n << " " << y.level << " " << y.scale << "\n";
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
