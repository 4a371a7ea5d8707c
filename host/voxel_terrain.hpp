// voxel_terrain.hpp -- C++ host-side mirror of the reference's chunk host API for the extraction
// path: PGRTerrain.Render.VoxelTerrain and the TerrainModifier interface
// (reference: Unity-Project/Assets/Scripts/VoxelTerrain.cs, TerrainModifier.cs, Utility.cs).
//
// Same names, argument meaning and error behaviour as the C# class, so a maintainer can diff them:
//   Init / InsertModifier / Update / Free          VoxelTerrain.cs:121, :251, :262, :214
//   BatchUpdate (private there, public here for tests) VoxelTerrain.cs:330-477
//   _width/_elevation/_height, _voxelScale, TerrainOrigin, blockSize, maxSampleResolution
// What differs on purpose: the three ComputeShader fields and the nine ComputeBuffer bindings
// (VoxelTerrain.cs:64-66, :370-421) are replaced by ONE vtmc context (include/vtmc.h); Unity
// objects (GameObject / Mesh / MeshCollider / Material, SetControlMap) have no equivalent here --
// a block's result is a plain BlockMesh {vertices, normals, triangles}.  No CPU extraction path
// exists: without libvtmc.so + a HIP device Init() throws.
#ifndef VTMC_HOST_VOXEL_TERRAIN_HPP
#define VTMC_HOST_VOXEL_TERRAIN_HPP

#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

struct vtmc_ctx;

namespace PGRTerrain {

// UnityEngine.Vector3 stand-in (only what the path uses)
struct Vector3 {
    float x = 0, y = 0, z = 0;
    Vector3() = default;
    Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vector3 operator+(const Vector3 &o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vector3 operator-(const Vector3 &o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vector3 operator*(float s) const { return {x * s, y * s, z * s}; }
    Vector3 operator/(float s) const { return {x / s, y / s, z / s}; }
    float sqrMagnitude() const { return x * x + y * y + z * z; }
    float magnitude() const;
    Vector3 normalized() const;
    static float Dot(const Vector3 &a, const Vector3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    static Vector3 ProjectOnPlane(const Vector3 &v, const Vector3 &n);
};
struct Vector2 {
    float x = 0, y = 0;
};

namespace MathHelper {
// Utility.cs:17-47 -- the block-index key
struct Int3 {
    int _x = 0, _y = 0, _z = 0;
    Int3() = default;
    Int3(int x, int y, int z) : _x(x), _y(y), _z(z) {}
    bool operator==(const Int3 &o) const { return _x == o._x && _y == o._y && _z == o._z; }
    int GetHashCode() const { return ((17 * 23 + _x) * 23 + _y) * 23 + _z; }  // Utility.cs:37-46
};
}  // namespace MathHelper

namespace Render {

// UnityException stand-in: thrown where the reference throws (VoxelTerrain.cs:123-142 ...)
class UnityException : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// What the GPU needs to evaluate a modifier itself (vtmc_modifier of include/vtmc.h, redeclared so
// this header stays free of the C ABI): kind 0 plane, 1 sphere, 2 cylinder + their parameters.
struct ModifierDesc {
    int kind = -1;
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float *data = nullptr;  // kind 3 (heightmap): float[dims[0]][dims[1]], owned by the modifier
    int dims[2] = {0, 0};
};

// TerrainModifier.cs:19-33
class TerrainModifier {
public:
    virtual ~TerrainModifier() = default;
    virtual Vector3 LowerBound() const = 0;
    virtual Vector3 UpperBound() const = 0;
    virtual float QueryDensity(const Vector3 &pos) const = 0;  // > 0 solid, < 0 air
    bool AddOrErode = true;                                    // true: union, false: difference
    // New: modifiers the library can evaluate on the device describe themselves; any other
    // (user-defined QueryDensity) returns false and is applied by the host loop as in the reference.
    virtual bool Describe(ModifierDesc &) const { return false; }
};

// TerrainModifier.cs:38-65  f = y0 - y
class PlaneModifier : public TerrainModifier {
public:
    float _height;
    Vector2 _low, _up;
    PlaneModifier(float height, Vector2 low, Vector2 up, bool addOrErode);
    Vector3 LowerBound() const override { return {_low.x, std::numeric_limits<float>::lowest(), _low.y}; }
    Vector3 UpperBound() const override { return {_up.x, _height + 1, _up.y}; }
    float QueryDensity(const Vector3 &pos) const override { return _height - pos.y; }
    bool Describe(ModifierDesc &d) const override
    {
        d.kind = 0;
        d.p[0] = _height;
        return true;
    }
};

// TerrainModifier.cs:70-91  f = r - |p - c|
class SphereModifier : public TerrainModifier {
public:
    Vector3 _center;
    float _radius;
    SphereModifier(Vector3 center, float radius, bool addOrErode) : _center(center), _radius(radius) { AddOrErode = addOrErode; }
    Vector3 LowerBound() const override { return {_center.x - _radius, _center.y - _radius, _center.z - _radius}; }
    Vector3 UpperBound() const override { return {_center.x + _radius, _center.y + _radius, _center.z + _radius}; }
    float QueryDensity(const Vector3 &pos) const override { return _radius - (pos - _center).magnitude(); }
    bool Describe(ModifierDesc &d) const override
    {
        d.kind = 1;
        d.p[0] = _center.x;
        d.p[1] = _center.y;
        d.p[2] = _center.z;
        d.p[3] = _radius;
        return true;
    }
};

// TerrainModifier.cs:96-152
class CylinderModifier : public TerrainModifier {
public:
    Vector3 _axisStart, _axisDir;
    float _axisLength, _radius;
    CylinderModifier(Vector3 start, Vector3 dir, float length, float radius, bool addOrErode);
    Vector3 LowerBound() const override;
    Vector3 UpperBound() const override;
    float QueryDensity(const Vector3 &pos) const override;
    bool Describe(ModifierDesc &d) const override
    {
        d.kind = 2;
        d.p[0] = _axisStart.x;
        d.p[1] = _axisStart.y;
        d.p[2] = _axisStart.z;
        d.p[3] = _axisDir.x;
        d.p[4] = _axisDir.y;
        d.p[5] = _axisDir.z;
        d.p[6] = _axisLength;
        d.p[7] = _radius;
        return true;
    }
};

// IslandModifier.cs:34-92: density = bilinear(_heightmap)(x, z) - y.  The reference fills _heightmap
// from Island.GetElevation (island generation, out of scope): this mirror is handed the array.
class IslandModifier : public TerrainModifier {
public:
    float _islandWidth, _islandHeight, _maxElevation;
    int _widthRes, _heightRes;
    std::vector<float> _heightmap;  // [u * _heightRes + v], as the C# float[widthRes, heightRes]
    IslandModifier(std::vector<float> heightmap, int widthRes, int heightRes, float islandWidth, float islandHeight,
                   float maxElevation, bool addOrErode = true);
    Vector3 LowerBound() const override { return {0, std::numeric_limits<float>::lowest(), 0}; }
    Vector3 UpperBound() const override { return {_islandWidth, _maxElevation, _islandHeight}; }
    float QueryDensity(const Vector3 &pos) const override;
    bool Describe(ModifierDesc &d) const override
    {
        d.kind = 3;
        d.p[0] = _islandWidth;
        d.p[1] = _islandHeight;
        d.data = _heightmap.data();
        d.dims[0] = _widthRes;
        d.dims[1] = _heightRes;
        return true;
    }
};

// What replaces a block's Unity Mesh (VoxelTerrain.cs:448-465): unindexed soup, indices 0..n-1.
struct BlockMesh {
    std::vector<Vector3> vertices;
    std::vector<Vector3> normals;
    std::vector<int> triangles;
    void Clear() { vertices.clear(); normals.clear(); triangles.clear(); }
};

// The 76-byte record of include/vtmc.h, redeclared so this header stays free of the C ABI.
struct CSTriangle {
    float _position0[3], _position1[3], _position2[3];
    float _normal0[3], _normal1[3], _normal2[3];
    int _block;
    static constexpr int stride = sizeof(float) * 3 * 6 + sizeof(int);  // VoxelTerrain.cs:36
};
static_assert(sizeof(CSTriangle) == 76, "CSTriangle must stay 76 bytes");

// The extraction seam of BatchUpdate (VoxelTerrain.cs:365-427).  The default implementation calls
// libvtmc.so; tests may install a recorder to check the host logic without a GPU.
struct ExtractBackend {
    virtual ~ExtractBackend() = default;
    // grid: float[(W+2),(E+2),(H+2)] z fastest; blocks: (x,y,z) triples.  Fills tris (canonical
    // order) and blockTriOffsets (B+1).  Throws UnityException on failure.
    virtual void Extract(const float *grid, int width, int elevation, int height, const std::vector<MathHelper::Int3> &blocks,
                         std::vector<CSTriangle> &tris, std::vector<int> &blockTriOffsets) = 0;

    // Device-resident terrain (vtmc_terrain_*): the grid lives in HBM, Update's density write runs on
    // the GPU.  A backend without it returns false from TerrainInit and the host path is used.
    struct QueuedModifier {
        ModifierDesc desc;
        bool addOrErode;
        Vector3 lower, upper;  // LowerBound / UpperBound, evaluated on the host as VoxelTerrain.cs:273-279 does
    };
    virtual bool TerrainInit(int, int, int, float, const Vector3 &, uint64_t) { return false; }
    // Applies the queue in order and extracts the dirty set; fills blocks (ordered by block id), tris, offsets.
    virtual void TerrainUpdate(const std::vector<QueuedModifier> &, std::vector<MathHelper::Int3> &, std::vector<CSTriangle> &,
                               std::vector<int> &)
    {
        throw std::logic_error("TerrainUpdate on a backend without device-resident terrain");
    }
    virtual void TerrainReadSamples(std::vector<float> &) { throw std::logic_error("TerrainReadSamples unsupported"); }
};

class VoxelTerrain {
public:
    // x: width, y: elevation, z: height (VoxelTerrain.cs:39-40)
    int _width = 16, _elevation = 16, _height = 16;
    static constexpr int maxSampleResolution = 1025;  // VoxelTerrain.cs:44
    static constexpr int blockSize = 8;               // VoxelTerrain.cs:54
    static constexpr int maxTriNumPerCell = 5;        // VoxelTerrain.cs:480
    float _voxelScale = 1.0f;                         // VoxelTerrain.cs:107
    Vector3 TerrainOrigin;                            // _transform.position, VoxelTerrain.cs:101
    int _device = 0;                                  // HIP device of the vtmc context (new)
    // New: keep _voxelSamples in HBM and run Update's density write on the GPU (vtmc_terrain_*).
    // Queues holding a modifier the device cannot evaluate (Describe() == false) are refused.
    bool _deviceResident = false;
    uint64_t _seed = 1;                               // seed of the device-side void / full values

    VoxelTerrain();
    ~VoxelTerrain();

    // Fresh random numbers on every read, VoxelTerrain.cs:50-51
    float voidDensity() { return _uniform(_rng) * 1.0f - 2.0f; }  // Random.Range(-2, -1)
    float fullDensity() { return _uniform(_rng) * 1.0f + 1.0f; }  // Random.Range(1, 2)
    Vector3 TerrainSize() const { return Vector3((float)_width, (float)_elevation, (float)_height) * _voxelScale; }

    void Init();                                                        // VoxelTerrain.cs:121-179
    void Free();                                                        // VoxelTerrain.cs:214-245
    void InsertModifier(std::shared_ptr<TerrainModifier> modifier);     // VoxelTerrain.cs:251-254
    void Update();                                                      // VoxelTerrain.cs:262-325
    void BatchUpdate();                                                 // VoxelTerrain.cs:330-477

    // -- inspection (tests, callers that consume the meshes) ----------------------------------
    const BlockMesh &Block(int x, int y, int z) const { return _blocks[((size_t)x * (_elevation / blockSize) + y) * (_height / blockSize) + z]; }
    float Sample(int x, int y, int z) const { return _voxelSamples[((size_t)x * (_elevation + 2) + y) * (_height + 2) + z]; }
    const std::vector<float> &Samples() const { return _voxelSamples; }
    std::vector<float> DeviceSamples() const;  // device-resident mode: the grid copied back (z fastest, like Samples())
    const std::vector<MathHelper::Int3> &LastUpdateBlocks() const { return _lastUpdateBlocks; }
    int LastTriangleCount() const { return _lastTriNum; }
    void SeedRandom(uint32_t seed) { _rng.seed(seed); }
    void SetBackend(std::shared_ptr<ExtractBackend> backend) { _backend = std::move(backend); }

private:
    void ApplyMeshes(const std::vector<CSTriangle> &csTriangles, const std::vector<int> &offsets);  // VoxelTerrain.cs:430-465
    std::vector<float> _voxelSamples;  // float[W+2, E+2, H+2], row-major, z fastest (VoxelTerrain.cs:145)
    std::vector<BlockMesh> _blocks;    // GameObject[,,] stand-in (VoxelTerrain.cs:61)
    std::vector<MathHelper::Int3> _nextUpdateblocks, _lastUpdateBlocks;
    std::deque<std::shared_ptr<TerrainModifier>> _modifierQueue;
    std::shared_ptr<ExtractBackend> _backend;
    bool _initialised = false;
    int _lastTriNum = 0;
    std::mt19937 _rng{12345u};
    std::uniform_real_distribution<float> _uniform{0.0f, 1.0f};
};

// Default backend: the HIP library behind include/vtmc.h.
std::shared_ptr<ExtractBackend> MakeVtmcBackend(int device);

}  // namespace Render
}  // namespace PGRTerrain
#endif
