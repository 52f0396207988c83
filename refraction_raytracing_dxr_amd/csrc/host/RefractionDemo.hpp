// RefractionDemo.hpp -- headless mirror of the reference's host API (RefractionDemo.hpp:9-10):
//   void initialize(HWND, int width, int height)   ->  int initialize(const Options&)
//   void drawFrame()                               ->  int drawFrame()
// The window handle is gone (no swap chain); the literals of RefractionDemo.cpp become Options
// whose defaults are those literals.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "Mesh.hpp"

namespace RefractionDemo {

struct Options {
    int width = 1024, height = 768;                       // WinMain.cpp:44
    std::string mesh_path = "../shell.obj";               // RefractionDemo.cpp:537
    std::string env_path = "../envMap.hdr";               // RefractionDemo.cpp:527
    int device = 0;
    float fov_y = (float)(52.0f / 180.0 * 3.1415);        // RefractionDemo.cpp:559
    float aspect = (float)1.333;
    float zn = 1.0f, zf = 125.0f;
    float angle0 = 0.01f, angle_step = 0.01f;             // RefractionDemo.cpp:555,567
    rr_dispatch_params dispatch;                          // RayTracing.hlsl literals
    Options() { rr_default_dispatch_params(&dispatch); }
};

int initialize(const Options& opt);       // RefractionDemo.cpp:513-553; returns rr_status
int drawFrame();                          // RefractionDemo.cpp:557-612; returns rr_status
// the frame loop as one call: no per-frame wait or read-back, `in_flight` launches overlapping (1..4)
int pump(int n_frames, int frames_per_dispatch, int in_flight, rr_stats* stats);
// the frame loop with every frame copied to host memory while the next ones render; frames: n_frames*w*h*4 bytes
int stream(int n_frames, int frames_per_dispatch, int in_flight, uint8_t* frames);
// Several GPUs, one process each (rank of world): frames shard by 32x32 tile, tile t to rank t % world; every rank renders
// its tiles of a batch of frames, ONE grouped RCCL gather brings them to rank 0, which de-interleaves them into whole frames.
// id128: the 128 bytes rank 0 got from rr_comm_unique_id, handed to every process.  The reference has a single adapter
// (RefractionDemo.cpp:163); this is the tile-parallel form of its frame loop.
int initializeSharded(const Options& opt, int rank, int world, const void* id128);
// n_frames of the orbit in batches of frames_per_gather; on rank 0 the last frame lands in backBuffer()
int pumpSharded(int n_frames, int frames_per_gather, rr_stats* stats);
// the frame drawFrame just produced (RGBA8, width*height*4), i.e. what Present would have shown
const std::vector<uint8_t>& backBuffer();
rr_context* context();
float currentAngle();
const char* lastError();
void shutdown();

} // namespace RefractionDemo
