// CPU-only checks of the host mirror (sfgwas_amd/host/gwas.hpp): GenoFileStream and DiagCacheStream semantics
// (gwas/filestream.go).  Usage: host_logic_test <workdir>   — prints "OK" or throws.
#include "../../sfgwas_amd/host/gwas.hpp"
#include <cassert>
#include <iostream>
#define REQUIRE(c) do { if (!(c)) { std::cerr << "FAILED: " #c " at line " << __LINE__ << std::endl; return 1; } } while (0)

int main(int argc, char **argv) {
    std::string dir = argc > 1 ? argv[1] : ".";
    // ---- GenoFileStream: 5 x 7 int8 matrix with missing values
    const int R = 5, Cc = 7;
    int8_t m[R][Cc];
    for (int i = 0; i < R; i++) for (int j = 0; j < Cc; j++) m[i][j] = (int8_t)(((i * 7 + j) % 4) - 1);   // -1,0,1,2
    { FILE *f = fopen((dir + "/g.bin").c_str(), "wb"); fwrite(m, 1, sizeof m, f); fclose(f); }
    {
        gwas::GenoFileStream gfs(dir + "/g.bin", R, Cc, true);
        REQUIRE(gfs.NumRows() == 5 && gfs.NumCols() == 7 && gfs.NumRowsToKeep() == 5 && gfs.NumColsToKeep() == 7);
        for (int i = 0; i < R; i++) { auto row = gfs.NextRow(); REQUIRE(row.size() == (size_t)Cc); for (int j = 0; j < Cc; j++) REQUIRE(row[j] == (m[i][j] < 0 ? 0 : m[i][j])); }
        REQUIRE(gfs.NextRow().empty());                         // EOF -> nil (filestream.go:415-417)
        gfs.Reset();
        REQUIRE(gfs.UpdateRowFilt({true, false, true, true, false}) == 3);
        REQUIRE(gfs.UpdateColFilt({true, true, false, true, false, true, true}) == 5);
        REQUIRE(gfs.NumRowsToKeep() == 3 && gfs.NumColsToKeep() == 5);
        REQUIRE(gfs.UpdateRowFilt({true, true, false}) == 2);   // filters compose over the kept entries (:428-454)
        bool threw = false; try { gfs.UpdateRowFilt({true}); } catch (const std::runtime_error &) { threw = true; }
        REQUIRE(threw);                                          // panic("Invalid length of input array")
        gfs.Reset();
        int keptRows[2] = {0, 2}, keptCols[5] = {0, 1, 3, 5, 6};
        for (int k = 0; k < 2; k++) { auto row = gfs.NextRow(); REQUIRE(row.size() == 5); for (int j = 0; j < 5; j++) { int8_t v = m[keptRows[k]][keptCols[j]]; REQUIRE(row[j] == (v < 0 ? 0 : v)); } }
    }
    {   // replaceMissing = false keeps -1
        gwas::GenoFileStream gfs(dir + "/g.bin", R, Cc, false);
        auto row = gfs.NextRow(); REQUIRE(row[0] == -1);
    }
    {   bool threw = false; try { gwas::GenoFileStream bad(dir + "/does_not_exist.bin", 1, 1, true); } catch (const std::runtime_error &) { threw = true; } REQUIRE(threw); }
    // ---- DiagCacheStream round trip in the reference's byte format
    const int slots = 16, d = 4, n = 8, nmod = 3;
    remove(gwas::DiagCacheStream::FileName(dir + "/cache", 0).c_str());
    {
        bool existed = false;
        gwas::DiagCacheStream w(dir + "/cache", 0, true, slots, &existed); REQUIRE(!existed);
        w.SetIndexTables({true, false, true, true}, {true, true, false, false});
        for (uint32_t shift : {0u, 2u, 7u}) {
            gwas::PlainVector pv; pv.pt.resize(3); pv.empty = {false, shift == 2, false};
            for (int k = 0; k < 3; k++) if (!pv.empty[k]) { pv.pt[k].resize(n * nmod); for (size_t x = 0; x < pv.pt[k].size(); x++) pv.pt[k][x] = 1000003ULL * (shift + 1) + 17 * k + x * 0x100000001ULL; }
            w.WriteDiag(pv, shift, 5, 17179869184.0, n, nmod);
        }
    }
    {
        bool existed = false;
        gwas::DiagCacheStream again(dir + "/cache", 0, true, slots, &existed); REQUIRE(existed);   // "Found cache file": skip (:52-54)
        gwas::DiagCacheStream r(dir + "/cache", 0, false, slots);
        REQUIRE(r.vectorLen == 3 && r.level == 5 && r.n == (uint64_t)n && r.numModuli == (uint64_t)nmod && r.scale == 17179869184.0);
        REQUIRE(r.rowSize == 4 + (1 + (uint64_t)n * nmod * 8) * 3);
        REQUIRE(r.babyTable[0] && !r.babyTable[1] && r.giantTable[1] && !r.giantTable[2]);
        gwas::PlainVector pv; int shift;
        for (uint32_t want : {0u, 2u, 7u}) {
            REQUIRE(r.ReadDiag(pv, shift)); REQUIRE((uint32_t)shift == want);
            REQUIRE(pv.empty[1] == (want == 2));
            REQUIRE(pv.pt[0][3] == 1000003ULL * (want + 1) + 3 * 0x100000001ULL);
        }
        REQUIRE(!r.ReadDiag(pv, shift));
    }
    // ---- cross-check with a file written by the oracle (argv[2]), if given: same header and payload decoding
    if (argc > 2) {
        gwas::DiagCacheStream r(argv[2], 0, false, slots);
        gwas::PlainVector pv; int shift; uint64_t acc = r.vectorLen * 1000 + r.level * 100 + r.numModuli;
        while (r.ReadDiag(pv, shift)) { acc = acc * 31 + (uint64_t)shift; for (size_t k = 0; k < pv.pt.size(); k++) if (!pv.empty[k]) for (uint64_t w : pv.pt[k]) acc = acc * 1099511628211ULL + w; }
        std::cout << "DIGEST " << acc << std::endl;
    }
    std::cout << "OK" << std::endl;
    return 0;
}
