#include "../../mcmc-db_amd/csrc/mcr_parquet.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
int main(int argc, char** argv)
{
    FILE* fp = fopen(argv[1], "rb"); if (!fp) return 2;
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    std::vector<unsigned char> img(n); if (fread(img.data(), 1, n, fp) != (size_t)n) return 2; fclose(fp);
    std::mt19937_64 rng(12345);
    unsigned flen; memcpy(&flen, img.data() + n - 8, 4);
    long lo = n - 8 - flen;
    int ok = 0, bad = 0;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    for (int it = 0; it < iters; ++it) {
        // exact-size heap copy so that any overrun is caught by ASAN
        std::vector<unsigned char> b(img);
        const int nm = 1 + rng() % 4;
        for (int k = 0; k < nm; ++k) {
            long pos;
            switch (rng() % 3) {
                case 0: pos = lo + rng() % (n - lo); break;                 // footer
                case 1: pos = rng() % 300; break;                           // first page headers
                default: pos = rng() % n;                                  // anywhere
            }
            b[pos] = (unsigned char)rng();
        }
        size_t len = b.size();
        if (rng() % 8 == 0) len = rng() % (len + 1);                        // truncation
        std::vector<unsigned char> c(b.begin(), b.begin() + len);
        mcr::pq::File f;
        if (mcr::pq::open(f, c.data(), c.size())) ++ok; else ++bad;
    }
    printf("opened %d rejected %d\n", ok, bad);
    return 0;
}
