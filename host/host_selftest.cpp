// host_selftest.cpp -- drives the C++ VoxelTerrain mirror the way TerrainEngine / SceneManager drive
// the C# class (TerrainEngine.cs:87,148,160; SceneManager.cs:121-129).
//   host_selftest --cpu            host logic only (recording backend, no GPU needed)
//   host_selftest --gpu <out_dir>  real libvtmc.so backend; dumps grid + meshes for the parity test
//   host_selftest --gpu-resident <out_dir>  the same scene, grid in HBM, Update on the device
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include "voxel_terrain.hpp"

using namespace PGRTerrain;
using namespace PGRTerrain::Render;
using MathHelper::Int3;

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

struct Recorder : ExtractBackend {
    int calls = 0;
    std::vector<Int3> blocks;
    std::vector<float> grid;
    void Extract(const float *g, int w, int e, int h, const std::vector<Int3> &b, std::vector<CSTriangle> &tris,
                 std::vector<int> &offs) override
    {
        ++calls;
        blocks = b;
        grid.assign(g, g + (size_t)(w + 2) * (e + 2) * (h + 2));
        tris.clear();
        offs.assign(b.size() + 1, 0);
    }
};

static bool throws_with(VoxelTerrain &vt, const char *needle)
{
    try {
        vt.Init();
    } catch (const UnityException &e) {
        return std::strstr(e.what(), needle) != nullptr;
    }
    return false;
}

static int cpu_tests()
{
    {  // VoxelTerrain.cs:138-142
        VoxelTerrain bad;
        bad.SetBackend(std::make_shared<Recorder>());
        bad._width = 10;
        CHECK(throws_with(bad, "block size must align to terrain size"));
        bad._width = 1032;
        CHECK(throws_with(bad, "too high resolution (exceeds 1025)"));
    }
    VoxelTerrain vt;
    auto rec = std::make_shared<Recorder>();
    vt.SetBackend(rec);
    vt._width = vt._elevation = vt._height = 64;
    vt.Init();
    for (float s : vt.Samples()) CHECK(s >= -2.0f && s <= -1.0f);  // voidDensity, VoxelTerrain.cs:145-149
    vt.Update();                                                  // empty queue: nothing happens
    CHECK(rec->calls == 0);

    vt.InsertModifier(std::make_shared<SphereModifier>(Vector3(16, 16, 16), 10.0f, true));
    vt.Update();
    CHECK(rec->calls == 1);
    // AABB [6,26] touches blocks 0..3 on each axis (inclusive test, VoxelTerrain.cs:311-313)
    CHECK(rec->blocks.size() == 64);
    for (const Int3 &b : rec->blocks) CHECK(b._x <= 3 && b._y <= 3 && b._z <= 3);
    float c = vt.Sample(16, 16, 16);
    CHECK(c >= 1.0f && c <= 2.0f);                        // clamp(10, void, full) = full in [1,2]
    float edge = vt.Sample(16, 16, 25);                   // r - 9 = 1 -> min(1, full) = 1
    CHECK(std::fabs(edge - 1.0f) < 1e-6f);
    float inside = vt.Sample(16, 16, 25 + 0) - vt.Sample(16, 16, 26);
    CHECK(inside > 0);                                    // density decreases outwards
    CHECK(vt.Sample(40, 40, 40) <= -1.0f);                // untouched samples stay void
    CHECK(std::fabs(vt.Sample(16, 22, 16) - 4.0f) > 1.9f);  // 10 - 6 = 4 is clamped into [1,2]

    // erode: S = clamp(min(S, -md)) (VoxelTerrain.cs:296-304)
    vt.InsertModifier(std::make_shared<SphereModifier>(Vector3(16, 16, 16), 4.0f, false));
    vt.Update();
    CHECK(rec->calls == 2);
    CHECK(vt.Sample(16, 16, 16) <= -1.0f);                // -clamp(4,..) = -full -> void range
    CHECK(rec->blocks.size() == 8);                       // AABB [12,20]: up >= 8b && low <= 8b+8 holds for b = 1, 2 only
    for (const Int3 &b : rec->blocks) CHECK(b._x >= 1 && b._x <= 2 && b._y >= 1 && b._y <= 2 && b._z >= 1 && b._z <= 2);
    return 0;
}

static int gpu_run(const std::string &out)
{
    VoxelTerrain vt;
    vt._width = 64;
    vt._elevation = 32;
    vt._height = 64;
    vt._voxelScale = 0.5f;
    vt.TerrainOrigin = Vector3(-3.0f, 1.0f, 2.0f);
    vt.SeedRandom(7);
    vt.Init();
    Vector2 lo, up;
    lo.x = -100; lo.y = -100; up.x = 100; up.y = 100;
    vt.InsertModifier(std::make_shared<PlaneModifier>(6.3f, lo, up, true));
    vt.InsertModifier(std::make_shared<SphereModifier>(Vector3(10.0f, 8.0f, 15.0f), 5.5f, true));
    vt.InsertModifier(std::make_shared<CylinderModifier>(Vector3(2.0f, 5.0f, 6.0f), Vector3(1.0f, 0.3f, 0.5f), 20.0f, 2.2f, false));
    vt.Update();
    std::printf("blocks %zu triangles %d\n", vt.LastUpdateBlocks().size(), vt.LastTriangleCount());

    std::ofstream(out + "/grid.f32", std::ios::binary)
        .write(reinterpret_cast<const char *>(vt.Samples().data()), (std::streamsize)(vt.Samples().size() * sizeof(float)));
    std::ofstream fb(out + "/blocks.i32", std::ios::binary), fv(out + "/vertices.f32", std::ios::binary),
        fn(out + "/normals.f32", std::ios::binary), fc(out + "/counts.i32", std::ios::binary);
    for (const Int3 &b : vt.LastUpdateBlocks()) {
        int xyz[3] = {b._x, b._y, b._z};
        fb.write(reinterpret_cast<const char *>(xyz), sizeof xyz);
        const BlockMesh &m = vt.Block(b._x, b._y, b._z);
        int n = (int)m.vertices.size();
        fc.write(reinterpret_cast<const char *>(&n), sizeof n);
        for (int i = 0; i < n; i++) {
            if (m.triangles[(size_t)i] != i) return 2;
            fv.write(reinterpret_cast<const char *>(&m.vertices[(size_t)i]), 12);
            fn.write(reinterpret_cast<const char *>(&m.normals[(size_t)i]), 12);
        }
    }
    // an edit like SceneManager.Update's mouse sphere (SceneManager.cs:121-129): small dirty set
    vt.InsertModifier(std::make_shared<SphereModifier>(Vector3(5.0f, 4.0f, 9.0f), 2.0f, false));
    vt.Update();
    std::printf("edit: blocks %zu triangles %d\n", vt.LastUpdateBlocks().size(), vt.LastTriangleCount());
    vt.Free();
    std::printf("HOST-GPU-OK\n");
    return 0;
}

// the same scene with the grid resident in HBM and Update's density write on the GPU
static int gpu_resident_run(const std::string &out)
{
    VoxelTerrain vt;
    vt._width = 64;
    vt._elevation = 32;
    vt._height = 64;
    vt._voxelScale = 0.5f;
    vt.TerrainOrigin = Vector3(-3.0f, 1.0f, 2.0f);
    vt._deviceResident = true;
    vt._seed = 4242;
    vt.Init();
    Vector2 lo, up;
    lo.x = -100; lo.y = -100; up.x = 100; up.y = 100;
    vt.InsertModifier(std::make_shared<PlaneModifier>(6.3f, lo, up, true));
    vt.InsertModifier(std::make_shared<SphereModifier>(Vector3(10.0f, 8.0f, 15.0f), 5.5f, true));
    vt.InsertModifier(std::make_shared<CylinderModifier>(Vector3(2.0f, 5.0f, 6.0f), Vector3(1.0f, 0.3f, 0.5f), 20.0f, 2.2f, false));
    {   // a small island on top (IslandModifier, the world-build modifier of TerrainEngine.cs:87)
        const int res = 9;
        std::vector<float> hm((size_t)res * res);
        for (int u = 0; u < res; u++)
            for (int v = 0; v < res; v++) hm[(size_t)u * res + v] = 4.0f + 0.5f * (float)((u * 3 + v * 5) % 7);
        vt.InsertModifier(std::make_shared<IslandModifier>(hm, res, res, 20.0f, 24.0f, 9.0f, true));
    }
    vt.Update();
    std::printf("resident: blocks %zu triangles %d\n", vt.LastUpdateBlocks().size(), vt.LastTriangleCount());
    vt.InsertModifier(std::make_shared<SphereModifier>(Vector3(5.0f, 4.0f, 9.0f), 2.0f, false));
    vt.Update();
    std::printf("resident edit: blocks %zu triangles %d\n", vt.LastUpdateBlocks().size(), vt.LastTriangleCount());
    const std::vector<float> grid = vt.DeviceSamples();
    std::ofstream(out + "/r_grid.f32", std::ios::binary).write(reinterpret_cast<const char *>(grid.data()), (std::streamsize)(grid.size() * sizeof(float)));
    std::ofstream fb(out + "/r_blocks.i32", std::ios::binary), fv(out + "/r_vertices.f32", std::ios::binary),
        fn(out + "/r_normals.f32", std::ios::binary), fc(out + "/r_counts.i32", std::ios::binary);
    for (const Int3 &b : vt.LastUpdateBlocks()) {
        int xyz[3] = {b._x, b._y, b._z};
        fb.write(reinterpret_cast<const char *>(xyz), sizeof xyz);
        const BlockMesh &m = vt.Block(b._x, b._y, b._z);
        int n = (int)m.vertices.size();
        fc.write(reinterpret_cast<const char *>(&n), sizeof n);
        for (int i = 0; i < n; i++) {
            fv.write(reinterpret_cast<const char *>(&m.vertices[(size_t)i]), 12);
            fn.write(reinterpret_cast<const char *>(&m.normals[(size_t)i]), 12);
        }
    }
    vt.Free();
    std::printf("HOST-RESIDENT-OK\n");
    return 0;
}

int main(int argc, char **argv)
{
    try {
        if (argc >= 2 && std::string(argv[1]) == "--cpu") {
            int rc = cpu_tests();
            if (rc == 0) std::printf("HOST-CPU-OK\n");
            return rc;
        }
        if (argc >= 3 && std::string(argv[1]) == "--gpu") return gpu_run(argv[2]);
        if (argc >= 3 && std::string(argv[1]) == "--gpu-resident") return gpu_resident_run(argv[2]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
    std::fprintf(stderr, "usage: host_selftest --cpu | --gpu <out_dir>\n");
    return 64;
}
