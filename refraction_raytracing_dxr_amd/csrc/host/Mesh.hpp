// Mesh.hpp -- host-side mirror of the reference's asset surface (Mesh.hpp:14-25), retargeted from
// D3D12 to the rrdxr C ABI.  Same public members and meanings:
//   bool load(const char*)              Mesh.cpp:6-37   (OBJ -> verts / indices)
//   raytracingGeometry() const          Mesh.cpp:39-53  (here: the ids the C ABI needs instead of a
//                                                        D3D12_RAYTRACING_GEOMETRY_DESC)
//   void upload(device)                 Mesh.cpp:55-94  (here: an rr_context instead of ID3D12Device5)
//   std::vector<uint32_t> indices; std::vector<Vertex> verts;
// draw() (Mesh.cpp:96-113) is the reference's dead raster path and has no counterpart.
#pragma once
#include <cstdint>
#include <vector>

#include "../../../include/rrdxr.h"

typedef rr_vertex Vertex;        // {position[3], norm[3], uv[2]}, 32 bytes

struct RaytracingGeometry {      // what BuildRaytracingAccelerationStructure needs to know
    uint32_t mesh_id;            // stands for the vertex/index buffer GPU addresses
    uint32_t vertex_count;
    uint32_t index_count;
    uint32_t vertex_stride;      // 32
};

struct Mesh {
    bool load(const char* filename);
    RaytracingGeometry raytracingGeometry() const;
    int  upload(rr_context* device);          // returns an rr_status (the reference returns void and ignores errors)

    std::vector<uint32_t> indices;
    std::vector<Vertex> verts;
    uint32_t mesh_id = 0xffffffffu;
};
