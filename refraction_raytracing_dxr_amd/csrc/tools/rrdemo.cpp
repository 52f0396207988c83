// rrdemo -- headless replacement of the reference's WinMain frame pump (WinMain.cpp:37-60):
// initialize once, drawFrame N times, optionally dump frames as binary PPM.
//   rrdemo --mesh shell.obj --env envmap.png [--size 1024x768] [--frames 10] [--out frame_%03d.ppm]
//          [--pump] [--frames-per-dispatch F] [--in-flight L]     (--pump: the loop without per-frame read-back)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../host/RefractionDemo.hpp"

int main(int argc, char** argv)
{
    RefractionDemo::Options opt;
    int frames = 1, fpd = 1, in_flight = 2;
    bool pump = false;
    std::string out;
    for (int i = 1; i < argc; ++i) {
        auto arg = [&](const char* name) { return !strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--mesh")) opt.mesh_path = argv[++i];
        else if (arg("--env")) opt.env_path = argv[++i];
        else if (arg("--frames")) frames = atoi(argv[++i]);
        else if (arg("--out")) out = argv[++i];
        else if (!strcmp(argv[i], "--pump")) pump = true;
        else if (arg("--frames-per-dispatch")) fpd = atoi(argv[++i]);
        else if (arg("--in-flight")) in_flight = atoi(argv[++i]);
        else if (arg("--device")) opt.device = atoi(argv[++i]);
        else if (arg("--max-refract")) opt.dispatch.max_refract = atoi(argv[++i]);
        else if (arg("--max-reflect")) opt.dispatch.max_reflect = atoi(argv[++i]);
        else if (arg("--size")) { if (sscanf(argv[++i], "%dx%d", &opt.width, &opt.height) != 2) { fprintf(stderr, "bad --size\n"); return 2; } }
        else { fprintf(stderr, "usage: rrdemo --mesh M.obj --env E.(hdr|png) [--size WxH] [--frames N] [--out f_%%03d.ppm]\n"); return 2; }
    }
    int rc = RefractionDemo::initialize(opt);
    if (rc != RR_OK) { fprintf(stderr, "initialize failed (%d): %s\n", rc, RefractionDemo::lastError()); return 1; }
    auto t0 = std::chrono::steady_clock::now();
    if (pump) {
        rr_stats st;
        if ((rc = RefractionDemo::pump(frames, fpd, in_flight, &st)) != RR_OK) { fprintf(stderr, "pump failed (%d): %s\n", rc, RefractionDemo::lastError()); return 1; }
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d frames of %dx%d in %.3f s (%.1f fps, %.1f Mrays/s; %d per dispatch, %d in flight)\n", frames, opt.width, opt.height, s,
               frames / s, (double)st.rays / s / 1e6, fpd, in_flight);
        if (!out.empty()) {                              // the frame the loop ended on
            char name[1024];
            snprintf(name, sizeof name, out.c_str(), frames - 1);
            FILE* f = fopen(name, "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", name); return 1; }
            fprintf(f, "P6\n%d %d\n255\n", opt.width, opt.height);
            const auto& bb = RefractionDemo::backBuffer();
            for (size_t p = 0; p < (size_t)opt.width * opt.height; ++p) fwrite(&bb[p * 4], 1, 3, f);
            fclose(f);
        }
        frames = 0;
    }
    for (int k = 0; k < frames; ++k) {
        if ((rc = RefractionDemo::drawFrame()) != RR_OK) { fprintf(stderr, "drawFrame failed (%d): %s\n", rc, RefractionDemo::lastError()); return 1; }
        if (!out.empty()) {
            char name[1024];
            snprintf(name, sizeof name, out.c_str(), k);
            FILE* f = fopen(name, "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", name); return 1; }
            fprintf(f, "P6\n%d %d\n255\n", opt.width, opt.height);
            const auto& bb = RefractionDemo::backBuffer();
            for (size_t p = 0; p < (size_t)opt.width * opt.height; ++p) fwrite(&bb[p * 4], 1, 3, f);
            fclose(f);
        }
    }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (frames) printf("%d frames of %dx%d in %.3f s (%.1f fps incl. readback)\n", frames, opt.width, opt.height, s, frames / s);
    RefractionDemo::shutdown();
    return 0;
}
