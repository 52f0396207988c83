// rrdemo -- headless replacement of the reference's WinMain frame pump (WinMain.cpp:37-60):
// initialize once, drawFrame N times, optionally dump frames as binary PPM.
//   rrdemo --mesh shell.obj --env envmap.png [--size 1024x768] [--frames 10] [--out frame_%03d.ppm]
//          [--pump] [--frames-per-dispatch F] [--in-flight L]     (--pump: the loop without per-frame read-back)
//          [--stream]                                              (every frame to host memory, copies overlap rendering)
//          [--gpus N [--frames-per-gather F]]                      (one process per GPU: tiles, one RCCL gather per F frames)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../host/RefractionDemo.hpp"

extern char** environ;

// --gpus N: this process only starts N copies of itself (ranks 0..N-1, device = --device + rank) and waits for them; it never
// touches a GPU.  Rank 0 makes the RCCL id and leaves it in a file the others wait for; the file lives in a directory only this
// user can enter (mkdtemp, mode 0700), so nobody else on the machine can plant it or a link in its place.  When a rank fails
// -- or cannot be started -- the others are ended too: they would wait in RCCL for it for ever.
static int launch_ranks(int argc, char** argv, int gpus)
{
    char dir[] = "/tmp/rrdemo.XXXXXX";
    if (!mkdtemp(dir)) { perror("mkdtemp"); return 1; }
    const std::string idfile = std::string(dir) + "/id";
    std::vector<pid_t> pids;
    bool ok = true;
    for (int r = 0; r < gpus && ok; ++r) {
        std::vector<std::string> a(argv, argv + argc);
        a.push_back("--rank"); a.push_back(std::to_string(r));
        a.push_back("--id-file"); a.push_back(idfile);
        std::vector<char*> av;
        for (auto& x : a) av.push_back(const_cast<char*>(x.c_str()));
        av.push_back(nullptr);
        pid_t pid;
        if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, av.data(), environ) != 0) { fprintf(stderr, "cannot start rank %d\n", r); ok = false; break; }
        pids.push_back(pid);
    }
    size_t left = pids.size();
    std::vector<bool> done(pids.size(), false);
    auto end_the_rest = [&] { for (size_t i = 0; i < pids.size(); ++i) if (!done[i]) kill(pids[i], SIGTERM); };
    if (!ok) end_the_rest();
    while (left > 0) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, 0);
        if (pid < 0) { ok = false; break; }
        for (size_t i = 0; i < pids.size(); ++i)
            if (pids[i] == pid && !done[i]) {
                done[i] = true; --left;
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { if (ok) end_the_rest(); ok = false; }
            }
    }
    unlink(idfile.c_str());
    unlink((idfile + ".tmp").c_str());
    rmdir(dir);
    return ok ? 0 : 1;
}

int main(int argc, char** argv)
{
    RefractionDemo::Options opt;
    int frames = 1, fpd = 1, in_flight = 2, gpus = 0, rank = -1, fpg = 16;
    bool pump = false, stream = false;
    std::string out, idfile;
    for (int i = 1; i < argc; ++i) {
        auto arg = [&](const char* name) { return !strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--mesh")) opt.mesh_path = argv[++i];
        else if (arg("--env")) opt.env_path = argv[++i];
        else if (arg("--frames")) frames = atoi(argv[++i]);
        else if (arg("--out")) out = argv[++i];
        else if (!strcmp(argv[i], "--pump")) pump = true;
        else if (!strcmp(argv[i], "--stream")) stream = true;
        else if (arg("--frames-per-dispatch")) fpd = atoi(argv[++i]);
        else if (arg("--in-flight")) in_flight = atoi(argv[++i]);
        else if (arg("--device")) opt.device = atoi(argv[++i]);
        else if (arg("--gpus")) gpus = atoi(argv[++i]);
        else if (arg("--frames-per-gather")) fpg = atoi(argv[++i]);
        else if (arg("--rank")) rank = atoi(argv[++i]);
        else if (arg("--id-file")) idfile = argv[++i];
        else if (!strcmp(argv[i], "--tonemap")) opt.dispatch.flags |= RR_DISPATCH_TONEMAP_REINHARD;      // c / (1 + c) before the UNORM8 store
        else if (arg("--max-refract")) opt.dispatch.max_refract = atoi(argv[++i]);
        else if (arg("--max-reflect")) opt.dispatch.max_reflect = atoi(argv[++i]);
        else if (arg("--size")) { if (sscanf(argv[++i], "%dx%d", &opt.width, &opt.height) != 2) { fprintf(stderr, "bad --size\n"); return 2; } }
        else { fprintf(stderr, "usage: rrdemo --mesh M.obj --env E.(hdr|png) [--size WxH] [--frames N] [--out f_%%03d.ppm]\n"); return 2; }
    }
    if (gpus > 0 && rank < 0) return launch_ranks(argc, argv, gpus);
    if (gpus > 0) {         // one rank of a sharded run
        unsigned char id[128];
        const std::string tmp = idfile + ".tmp";
        if (rank == 0) {
            if (rr_comm_unique_id(id) != RR_OK) { fprintf(stderr, "rr_comm_unique_id failed (is librccl.so installed?)\n"); return 1; }
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
            fclose(f);
            rename(tmp.c_str(), idfile.c_str());
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 6000 && !(f = fopen(idfile.c_str(), "rb")); ++tries) usleep(10000);
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "rank %d: no RCCL id from rank 0\n", rank); return 1; }
            fclose(f);
        }
        opt.device += rank;
        int rc = RefractionDemo::initializeSharded(opt, rank, gpus, id);
        if (rc != RR_OK) { fprintf(stderr, "rank %d: initialize failed (%d): %s\n", rank, rc, RefractionDemo::lastError()); return 1; }
        auto t0 = std::chrono::steady_clock::now();
        rr_stats st;
        if ((rc = RefractionDemo::pumpSharded(frames, fpg, &st)) != RR_OK) { fprintf(stderr, "rank %d: pump failed (%d): %s\n", rank, rc, RefractionDemo::lastError()); return 1; }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("rank %d of %d: %d frames of %dx%d in %.3f s (%.1f fps; %.1f Mrays/s on this rank; %d frames per gather)\n", rank, gpus, frames,
               opt.width, opt.height, s, frames / s, (double)st.rays / s / 1e6, fpg);
        if (rank == 0 && !out.empty()) {
            char name[1024];
            snprintf(name, sizeof name, out.c_str(), frames - 1);
            FILE* f = fopen(name, "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", name); return 1; }
            fprintf(f, "P6\n%d %d\n255\n", opt.width, opt.height);
            const auto& bb = RefractionDemo::backBuffer();
            for (size_t p = 0; p < (size_t)opt.width * opt.height; ++p) fwrite(&bb[p * 4], 1, 3, f);
            fclose(f);
        }
        RefractionDemo::shutdown();
        return 0;
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
    if (stream && frames > 0) {
        // chunks of up to 64 frames through one page-locked host buffer; --out writes every frame of every chunk
        const size_t fb = (size_t)opt.width * opt.height * 4;
        const int chunk = frames < 64 ? frames : 64;
        std::vector<uint8_t> host((size_t)chunk * fb);
        const bool pinned = rr_host_register(RefractionDemo::context(), host.data(), host.size()) == RR_OK;
        t0 = std::chrono::steady_clock::now();                      // page-locking the buffer is set-up, not frame time
        double first_chunk_s = 0.0;
        for (int k0 = 0; k0 < frames; k0 += chunk) {
            const int n = frames - k0 < chunk ? frames - k0 : chunk;
            if (k0 == chunk) first_chunk_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if ((rc = RefractionDemo::stream(n, fpd, in_flight, host.data())) != RR_OK) { fprintf(stderr, "stream failed (%d): %s\n", rc, RefractionDemo::lastError()); return 1; }
            for (int k = 0; k < n && !out.empty(); ++k) {
                char name[1024];
                snprintf(name, sizeof name, out.c_str(), k0 + k);
                FILE* f = fopen(name, "wb");
                if (!f) { fprintf(stderr, "cannot write %s\n", name); return 1; }
                fprintf(f, "P6\n%d %d\n255\n", opt.width, opt.height);
                for (size_t p = 0; p < (size_t)opt.width * opt.height; ++p) fwrite(&host[(size_t)k * fb + p * 4], 1, 3, f);
                fclose(f);
            }
        }
        if (pinned) rr_host_unregister(RefractionDemo::context(), host.data());
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d frames of %dx%d in %.3f s (%.1f fps delivered to %s host memory; %d per dispatch, %d in flight)\n", frames, opt.width,
               opt.height, s, frames / s, pinned ? "page-locked" : "pageable", fpd, in_flight);
        if (first_chunk_s > 0.0 && out.empty())     // the first chunk pays for cold code, streams and first-touch of the host pages
            printf("after the first %d frames: %.1f fps\n", chunk, (frames - chunk) / (s - first_chunk_s));
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
