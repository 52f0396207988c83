#include "Mesh.hpp"

#include <cstring>

bool Mesh::load(const char* filename)
{
    rr_vertex* v = nullptr;
    uint32_t* i = nullptr;
    uint32_t nv = 0, ni = 0;
    if (rr_host_mesh_load_obj(filename, &v, &nv, &i, &ni) != RR_OK) return false;
    // the reference appends to whatever the vectors already hold (Mesh.cpp:31-32)
    const uint32_t base = (uint32_t)verts.size();
    verts.insert(verts.end(), v, v + nv);
    for (uint32_t k = 0; k < ni; ++k) indices.push_back(base + i[k]);
    rr_host_free(v);
    rr_host_free(i);
    return true;
}

RaytracingGeometry Mesh::raytracingGeometry() const
{
    RaytracingGeometry g;
    g.mesh_id = mesh_id;
    g.vertex_count = (uint32_t)verts.size();
    g.index_count = (uint32_t)indices.size();
    g.vertex_stride = (uint32_t)sizeof(Vertex);
    return g;
}

int Mesh::upload(rr_context* device)
{
    return rr_upload_mesh(device, verts.data(), (uint32_t)verts.size(), indices.data(), (uint32_t)indices.size(), &mesh_id);
}
