#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
extern "C" int al3d_jpeg_header(const unsigned char*, int64_t, int*, unsigned short*);
extern "C" int al3d_jpeg_entropy_decode(const unsigned char*, int64_t, short*, int64_t);
thread_local char g_al3d_err[512];
int main(int argc, char** argv)
{
    int okc = 0, errc = 0;
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb"); if (!f) continue;
        std::vector<unsigned char> base; unsigned char buf[65536]; size_t r;
        while ((r = fread(buf, 1, sizeof buf, f)) > 0) base.insert(base.end(), buf, buf + r);
        fclose(f);
        uint64_t rng = 0x9e3779b97f4a7c15ull + a;
        auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        for (int it = 0; it < 4000; ++it) {
            std::vector<unsigned char> d = base;
            const int mode = it % 4;
            if (mode == 0) { const int flips = 1 + next() % 8; for (int k = 0; k < flips; ++k) d[next() % d.size()] ^= (unsigned char)(1u << (next() % 8)); }
            else if (mode == 1) { d.resize(1 + next() % d.size()); }
            else if (mode == 2) { const size_t hdr = d.size() < 700 ? d.size() : 700; const int n = 1 + next() % 6; for (int k = 0; k < n; ++k) d[next() % hdr] = (unsigned char)next(); }
            else { const size_t p = next() % d.size(); const size_t n = 1 + next() % 64; for (size_t k = 0; k < n && p + k < d.size(); ++k) d[p + k] = (unsigned char)next(); }
            // exact-size heap copy so that ASan sees any overread
            unsigned char* h = (unsigned char*)malloc(d.size()); memcpy(h, d.data(), d.size());
            int info[32]; unsigned short q[192];
            int rc = al3d_jpeg_header(h, (int64_t)d.size(), info, q);
            if (rc == 0) {
                const int64_t blocks = info[20];
                if (blocks > 0 && blocks < (1 << 22)) {
                    short* co = (short*)malloc((size_t)blocks * 128);
                    rc = al3d_jpeg_entropy_decode(h, (int64_t)d.size(), co, blocks);
                    free(co);
                }
            }
            if (rc == 0) ++okc; else ++errc;
            free(h);
        }
    }
    printf("fuzz: %d decoded, %d refused, no crash\n", okc, errc);
    return 0;
}
