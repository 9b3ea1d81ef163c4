// GPU check of the host mirror: MatMult4Stream / MatMult4StreamPreprocess+Compute / RotateRight called exactly the
// way the Go callers call them (assoc.go:395, pca.go:112-113,344,352), on files prepared by tests/test_host_mirror.py.
// Usage: host_gpu_test <casedir>.  Inputs (little-endian): moduli.bin (nq, np, u64...), keys.bin (count, then per key:
// galois, words...), A.bin, geno.bin + dims in case.txt.  rlk.bin.  Outputs: out_stream.bin, sums.bin, out_xt.bin, rot.bin, cmult.bin, csub.bin, masktrunc.bin, cmultconst.bin.
#include "../../sfgwas_amd/host/gwas.hpp"
#include <fstream>
#include <iostream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
static void writeU64(const std::string &fn, const std::vector<uint64_t> &v) { std::ofstream f(fn, std::ios::binary); f.write((const char *)v.data(), v.size() * 8); }
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::ifstream cs(dir + "/case.txt"); uint64_t nrow, ncol; int s, level, maxLevel, square; cs >> nrow >> ncol >> s >> level >> maxLevel >> square;
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, 17179869184.0);
        const int N = cps->N(), beta = (nq + np - 1) / np; const size_t kw = (size_t)beta * 2 * (nq + np) * N;
        auto keys = readU64(dir + "/keys.bin");
        for (size_t k = 0, off = 1; k < keys[0]; k++, off += 1 + kw) crypto::LoadRotationKey(cps.get(), keys[off], std::vector<uint64_t>(keys.begin() + off + 1, keys.begin() + off + 1 + kw), false);
        const int slots = cps->GetSlots(), nbr = (int)((nrow - 1) / slots) + 1, m_ct = (int)((ncol - 1) / slots) + 1;
        auto aflat = readU64(dir + "/A.bin");
        crypto::CipherMatrix A = gwas::unflatten(aflat, s, nbr, level, 17179869184.0, N);
        // --- association-style call (assoc.go:424)
        gwas::GenoFileStream gfs(dir + "/geno.bin", nrow, ncol, false);
        auto [out, sum, sq] = gwas::MatMult4Stream(cps.get(), A, &gfs, maxLevel, true, square != 0, 0);
        if ((int)out.size() != s || (int)out[0].size() != m_ct || out[0][0].Level() != maxLevel - 1) throw std::runtime_error("bad output shape");
        writeU64(dir + "/out_stream.bin", gwas::flattenCipherMatrix(out));
        std::vector<uint64_t> sb(2 * ncol); memcpy(sb.data(), sum.data(), ncol * 8); memcpy(sb.data() + ncol, sq.data(), ncol * 8); writeU64(dir + "/sums.bin", sb);
        // --- PCA-style calls (pca.go:112-113, matmult.go:42,91): one resident copy serves X and X^T
        gwas::GenoFileStream gfs2(dir + "/geno.bin", nrow, ncol, true);
        gwas::MatMult4StreamPreprocess(cps.get(), &gfs2, maxLevel, dir + "/cache_X");
        gwas::MatMult4StreamPreprocess(cps.get(), nullptr, maxLevel, dir + "/cache_XT", dir + "/cache_X");
        if (!square) {
            auto o1 = gwas::MatMult4StreamCompute(cps.get(), A, maxLevel, dir + "/cache_X", m_ct);
            if (gwas::flattenCipherMatrix(o1) != gwas::flattenCipherMatrix(out)) throw std::runtime_error("Compute(X) differs from MatMult4Stream");
        }
        auto atflat = readU64(dir + "/AT.bin");
        crypto::CipherMatrix AT = gwas::unflatten(atflat, s, m_ct, level, 17179869184.0, N);
        auto o2 = gwas::MatMult4StreamCompute(cps.get(), AT, maxLevel, dir + "/cache_XT", nbr);
        writeU64(dir + "/out_xt.bin", gwas::flattenCipherMatrix(o2));
        // --- crypto.RotateRight (basics.go:212)
        auto r = crypto::RotateRight(cps.get(), A[0][0], 1);
        writeU64(dir + "/rot.bin", r.data);
        // --- crypto.CMult / CSub (basics.go:386,580) the way qrfact / pca call them between the products
        crypto::LoadRelinKey(cps.get(), readU64(dir + "/rlk.bin"), false);
        auto prod = crypto::CMult(cps.get(), A[0], A[1], qi);
        if (prod[0].level != level - 1) throw std::runtime_error("CMult must rescale exactly once at scale 2^68");
        writeU64(dir + "/cmult.bin", prod[0].data);
        writeU64(dir + "/csub.bin", crypto::CSub(cps.get(), A[0], A[1])[0].data);
        // --- crypto.CMultConstRescale (basics.go:533; pca.go / qrfact.go scale by 1/n style constants)
        auto cm = crypto::CMultConstRescale(cps.get(), A[0], 1.0 / 8192.0, qi);
        if (cm[0].level != level - 1) throw std::runtime_error("CMultConstRescale must consume one level for a fractional constant");
        writeU64(dir + "/cmultconst.bin", cm[0].data);
        // --- crypto.MaskTrunc (basics.go:110) as QXLazyNormStream applies it to the tail ciphertext (matmult.go:62-70)
        auto mt = crypto::MaskTrunc(cps.get(), A[0][0], 1000, qi);
        if (mt.level != level - 1) throw std::runtime_error("MaskTrunc must consume one level");
        writeU64(dir + "/masktrunc.bin", mt.data);
        bool threw = false; try { gwas::MatMult4StreamCompute(cps.get(), A, maxLevel, dir + "/no_such_cache", m_ct); } catch (const std::runtime_error &) { threw = true; }
        if (!threw) throw std::runtime_error("missing cache prefix must fail loudly");
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
