// Sanitizer driver for the pure-host part of the C ABI (OBJ loader, PNG / Radiance decoder): built with
// g++ -fsanitize=address,undefined by tests/test_host_sanitized.py and fed valid, truncated and bit-flipped files.
// Any memory error or undefined behaviour aborts with a non-zero exit code; a clean refusal is fine.
//   host_san_driver img <file>... | obj <file>... | objx <file>...
#include <cstdio>
#include <cstring>
#include "../../include/rrdxr.h"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    unsigned ok = 0, refused = 0;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[1], "img")) {
            for (int req = 0; req <= 4; ++req) {
                int x = 0, y = 0, n = 0;
                float* p = rr_host_image_loadf(argv[i], &x, &y, &n, req);
                if (p) {
                    const int c = req ? req : n;
                    volatile float s = 0;                           // touch every value the decoder claims to have produced
                    for (long k = 0; k < (long)x * y * c; ++k) s = s + p[k];
                    rr_host_free(p);
                    ++ok;
                } else ++refused;
            }
        } else {
            rr_vertex* v = nullptr; uint32_t* idx = nullptr; uint32_t nv = 0, ni = 0;
            const int rc = !strcmp(argv[1], "objx") ? rr_host_mesh_load_obj_ex(argv[i], RR_OBJ_HARDENED, &v, &nv, &idx, &ni)
                                                    : rr_host_mesh_load_obj(argv[i], &v, &nv, &idx, &ni);
            if (rc == RR_OK) {
                volatile float s = 0;
                for (uint32_t k = 0; k < ni; ++k) s = s + v[idx[k]].position[0] + v[idx[k]].norm[2] + v[idx[k]].uv[1];
                rr_host_free(v); rr_host_free(idx);
                ++ok;
            } else ++refused;
        }
    }
    printf("ok %u refused %u\n", ok, refused);
    return 0;
}
