// voxel_terrain.cpp -- see voxel_terrain.hpp.  Reference lines are cited per function
// (Unity-Project/Assets/Scripts/VoxelTerrain.cs unless stated otherwise).
#include "voxel_terrain.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_set>

#include "../include/vtmc.h"

namespace PGRTerrain {

float Vector3::magnitude() const { return std::sqrt(x * x + y * y + z * z); }
Vector3 Vector3::normalized() const
{
    float m = magnitude();
    return m > 1e-5f ? *this / m : Vector3();  // Unity returns zero for tiny vectors
}
Vector3 Vector3::ProjectOnPlane(const Vector3 &v, const Vector3 &n)
{
    float d = Dot(n, n);
    if (d < 1e-12f) return v;
    return v - n * (Dot(v, n) / d);
}

namespace Render {

using MathHelper::Int3;

static float Clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }  // Mathf.Clamp

PlaneModifier::PlaneModifier(float height, Vector2 low, Vector2 up, bool addOrErode) : _height(height), _low(low), _up(up)
{
    if (low.x > up.x || low.y > up.y) throw UnityException("invalud aabb");  // TerrainModifier.cs:52 (sic)
    AddOrErode = addOrErode;
}

CylinderModifier::CylinderModifier(Vector3 start, Vector3 dir, float length, float radius, bool addOrErode)
    : _axisStart(start), _axisDir(dir.normalized()), _axisLength(length), _radius(radius)
{
    AddOrErode = addOrErode;
}

// TerrainModifier.cs:103-116
Vector3 CylinderModifier::LowerBound() const
{
    Vector3 leftProj = Vector3::ProjectOnPlane({-1, 0, 0}, _axisDir);
    Vector3 downProj = Vector3::ProjectOnPlane({0, -1, 0}, _axisDir);
    Vector3 backProj = Vector3::ProjectOnPlane({0, 0, -1}, _axisDir);
    Vector3 end = _axisStart + _axisDir * _axisLength;
    return {_axisDir.x > 0 ? (_axisStart + leftProj * _radius).x : (end + leftProj * _radius).x,
            _axisDir.y > 0 ? (_axisStart + downProj * _radius).y : (end + downProj * _radius).y,
            _axisDir.z > 0 ? (_axisStart + backProj * _radius).z : (end + backProj * _radius).z};
}

// TerrainModifier.cs:118-131
Vector3 CylinderModifier::UpperBound() const
{
    Vector3 rightProj = Vector3::ProjectOnPlane({1, 0, 0}, _axisDir);
    Vector3 upProj = Vector3::ProjectOnPlane({0, 1, 0}, _axisDir);
    Vector3 foreProj = Vector3::ProjectOnPlane({0, 0, 1}, _axisDir);
    Vector3 end = _axisStart + _axisDir * _axisLength;
    return {_axisDir.x < 0 ? (_axisStart + rightProj * _radius).x : (end + rightProj * _radius).x,
            _axisDir.y < 0 ? (_axisStart + upProj * _radius).y : (end + upProj * _radius).y,
            _axisDir.z < 0 ? (_axisStart + foreProj * _radius).z : (end + foreProj * _radius).z};
}

// TerrainModifier.cs:143-149
float CylinderModifier::QueryDensity(const Vector3 &pos) const
{
    Vector3 start2pos = pos - _axisStart;
    float projLength = Vector3::Dot(start2pos, _axisDir);
    return std::min({projLength, _axisLength - projLength,
                     _radius - std::sqrt(start2pos.sqrMagnitude() - projLength * projLength)});
}

IslandModifier::IslandModifier(std::vector<float> heightmap, int widthRes, int heightRes, float islandWidth, float islandHeight,
                               float maxElevation, bool addOrErode)
    : _islandWidth(islandWidth), _islandHeight(islandHeight), _maxElevation(maxElevation), _widthRes(widthRes), _heightRes(heightRes),
      _heightmap(std::move(heightmap))
{
    if (widthRes < 1 || heightRes < 1 || _heightmap.size() != (size_t)widthRes * heightRes) throw UnityException("heightmap size mismatch");
    AddOrErode = addOrErode;
}

// IslandModifier.cs:45-73
float IslandModifier::QueryDensity(const Vector3 &pos) const
{
    auto lerp = [](float a, float b, float t) { return a + (b - a) * Clamp(t, 0.0f, 1.0f); };  // Mathf.Lerp
    float u = Clamp(pos.x, 0.0f, _islandWidth);
    u = u / _islandWidth * (float)(_widthRes - 1);
    u = Clamp(u, 0.0f, (float)(_widthRes - 1));
    float v = Clamp(pos.z, 0.0f, _islandHeight);
    v = v / _islandHeight * (float)(_heightRes - 1);
    v = Clamp(v, 0.0f, (float)(_heightRes - 1));
    const int u0 = (int)std::floor(u), u1 = (int)std::ceil(u), v0 = (int)std::floor(v), v1 = (int)std::ceil(v);
    const float h00 = _heightmap[(size_t)u0 * _heightRes + v0], h10 = _heightmap[(size_t)u1 * _heightRes + v0];
    const float h01 = _heightmap[(size_t)u0 * _heightRes + v1], h11 = _heightmap[(size_t)u1 * _heightRes + v1];
    const float h0 = lerp(h00, h01, v - (float)v0), h1 = lerp(h10, h11, v - (float)v0);
    return lerp(h0, h1, u - (float)u0) - pos.y;
}

// ---------------------------------------------------------------------------------------------
// default backend: libvtmc.so
// ---------------------------------------------------------------------------------------------
namespace {
class VtmcBackend : public ExtractBackend {
public:
    explicit VtmcBackend(int device)
    {
        // replaces the three table uploads of Init (VoxelTerrain.cs:151-156)
        if (vtmc_create(device, &_ctx) != VTMC_OK) throw UnityException(std::string("vtmc_create: ") + vtmc_last_error(nullptr));
    }
    ~VtmcBackend() override { vtmc_destroy(_ctx); }  // VoxelTerrain.cs:228-244
    void Extract(const float *grid, int w, int e, int h, const std::vector<Int3> &blocks, std::vector<CSTriangle> &tris,
                 std::vector<int> &offsets) override
    {
        std::vector<int32_t> list;
        list.reserve(blocks.size() * 3);
        for (const Int3 &b : blocks) {
            list.push_back(b._x);
            list.push_back(b._y);
            list.push_back(b._z);
        }
        int32_t triNum = 0;
        // a C# float[W+2,E+2,H+2] is z fastest: strides ((E+2)(H+2), H+2, 1)
        int rc = vtmc_extract_grid(_ctx, grid, w, e, h, (int64_t)(e + 2) * (h + 2), h + 2, 1, list.data(), (int32_t)blocks.size(), &triNum);
        if (rc != VTMC_OK) throw UnityException(std::string("vtmc_extract_grid: ") + vtmc_last_error(_ctx));
        tris.resize((size_t)triNum);
        offsets.assign(blocks.size() + 1, 0);
        rc = vtmc_read_triangles(_ctx, reinterpret_cast<vtmc_triangle *>(tris.data()), triNum, offsets.data());
        if (rc != VTMC_OK) throw UnityException(std::string("vtmc_read_triangles: ") + vtmc_last_error(_ctx));
    }

    bool TerrainInit(int w, int e, int h, float scale, const Vector3 &origin, uint64_t seed) override
    {
        const float o[3] = {origin.x, origin.y, origin.z};
        if (vtmc_terrain_init(_ctx, w, e, h, scale, o, seed) != VTMC_OK)
            throw UnityException(std::string("vtmc_terrain_init: ") + vtmc_last_error(_ctx));
        _dims[0] = w;
        _dims[1] = e;
        _dims[2] = h;
        return true;
    }
    void TerrainUpdate(const std::vector<QueuedModifier> &queue, std::vector<Int3> &blocks, std::vector<CSTriangle> &tris,
                       std::vector<int> &offsets) override
    {
        std::vector<vtmc_modifier> mods(queue.size());
        for (size_t i = 0; i < queue.size(); i++) {
            vtmc_modifier &m = mods[i];
            m.kind = queue[i].desc.kind;
            m.add_or_erode = queue[i].addOrErode ? 1 : 0;
            const Vector3 &lo = queue[i].lower, &up = queue[i].upper;
            m.lower[0] = lo.x, m.lower[1] = lo.y, m.lower[2] = lo.z;
            m.upper[0] = up.x, m.upper[1] = up.y, m.upper[2] = up.z;
            std::memcpy(m.p, queue[i].desc.p, sizeof m.p);
            m.data = queue[i].desc.data;
            m.data_dims[0] = queue[i].desc.dims[0];
            m.data_dims[1] = queue[i].desc.dims[1];
        }
        int32_t nDirty = 0, triNum = 0;
        if (vtmc_terrain_update(_ctx, mods.data(), (int32_t)mods.size(), &nDirty, &triNum) != VTMC_OK)
            throw UnityException(std::string("vtmc_terrain_update: ") + vtmc_last_error(_ctx));
        std::vector<int32_t> list((size_t)nDirty * 3);
        if (vtmc_terrain_dirty_blocks(_ctx, list.data(), nDirty, nullptr) != VTMC_OK)
            throw UnityException(std::string("vtmc_terrain_dirty_blocks: ") + vtmc_last_error(_ctx));
        blocks.clear();
        for (int32_t i = 0; i < nDirty; i++) blocks.emplace_back(list[3 * (size_t)i], list[3 * (size_t)i + 1], list[3 * (size_t)i + 2]);
        tris.resize((size_t)triNum);
        offsets.assign((size_t)nDirty + 1, 0);
        if (nDirty > 0 && vtmc_read_triangles(_ctx, reinterpret_cast<vtmc_triangle *>(tris.data()), triNum, offsets.data()) != VTMC_OK)
            throw UnityException(std::string("vtmc_read_triangles: ") + vtmc_last_error(_ctx));
    }
    void TerrainReadSamples(std::vector<float> &out) override
    {
        const int64_t ey = _dims[1] + 2, ez = _dims[2] + 2;
        out.resize((size_t)(_dims[0] + 2) * ey * ez);
        // into the C# float[W+2,E+2,H+2] layout: z fastest
        if (vtmc_terrain_read_samples(_ctx, out.data(), ey * ez, ez, 1) != VTMC_OK)
            throw UnityException(std::string("vtmc_terrain_read_samples: ") + vtmc_last_error(_ctx));
    }

private:
    vtmc_ctx *_ctx = nullptr;
    int _dims[3] = {0, 0, 0};
};
}  // namespace

std::shared_ptr<ExtractBackend> MakeVtmcBackend(int device) { return std::make_shared<VtmcBackend>(device); }

// ---------------------------------------------------------------------------------------------
VoxelTerrain::VoxelTerrain() = default;
VoxelTerrain::~VoxelTerrain() = default;

// VoxelTerrain.cs:121-179
void VoxelTerrain::Init()
{
    if (_width % blockSize != 0 || _elevation % blockSize != 0 || _height % blockSize != 0)
        throw UnityException("block size must align to terrain size");
    if (_width + 1 > maxSampleResolution || _elevation + 1 > maxSampleResolution || _height + 1 > maxSampleResolution)
        throw UnityException("too high resolution (exceeds " + std::to_string(maxSampleResolution) + ")");
    if (_width <= 0 || _elevation <= 0 || _height <= 0) throw UnityException("block size must align to terrain size");

    _blocks.assign((size_t)(_width / blockSize) * (_elevation / blockSize) * (_height / blockSize), BlockMesh());
    // augmented by one layer so normals on the positive boundary are defined (VoxelTerrain.cs:145)
    if (!_backend) _backend = MakeVtmcBackend(_device);  // throws when no HIP device: there is no CPU path
    if (_deviceResident) {
        _voxelSamples.clear();  // the grid lives in HBM
        if (!_backend->TerrainInit(_width, _elevation, _height, _voxelScale, TerrainOrigin, _seed))
            throw UnityException("backend has no device-resident terrain");
    } else {
        _voxelSamples.resize((size_t)(_width + 2) * (_elevation + 2) * (_height + 2));
        for (float &s : _voxelSamples) s = voidDensity();
    }
    _nextUpdateblocks.clear();
    _modifierQueue.clear();
    _initialised = true;
}

// VoxelTerrain.cs:214-245
void VoxelTerrain::Free()
{
    for (BlockMesh &b : _blocks) b.Clear();
    _blocks.clear();
    _backend.reset();
    _initialised = false;
}

// VoxelTerrain.cs:251-254
void VoxelTerrain::InsertModifier(std::shared_ptr<TerrainModifier> modifier) { _modifierQueue.push_back(std::move(modifier)); }

// VoxelTerrain.cs:262-325
void VoxelTerrain::Update()
{
    if (!_initialised) throw UnityException("VoxelTerrain.Update before Init");
    if (_deviceResident) {
        // the whole of Update on the device: density writes, dirty set, BatchUpdate (VoxelTerrain.cs:262-325)
        std::vector<ExtractBackend::QueuedModifier> queue;
        std::vector<std::shared_ptr<TerrainModifier>> alive;  // descriptors borrow from the modifiers (heightmap) until the call returns
        while (!_modifierQueue.empty()) {
            std::shared_ptr<TerrainModifier> modifier = _modifierQueue.front();
            _modifierQueue.pop_front();
            ExtractBackend::QueuedModifier q;
            if (!modifier->Describe(q.desc)) throw UnityException("modifier cannot be evaluated on the device (Describe() returned false)");
            q.addOrErode = modifier->AddOrErode;
            q.lower = modifier->LowerBound();
            q.upper = modifier->UpperBound();
            queue.push_back(q);
            alive.push_back(std::move(modifier));
        }
        _lastUpdateBlocks.clear();
        _lastTriNum = 0;
        if (queue.empty()) return;
        std::vector<CSTriangle> csTriangles;
        std::vector<int> offsets;
        _backend->TerrainUpdate(queue, _nextUpdateblocks, csTriangles, offsets);
        _lastUpdateBlocks = _nextUpdateblocks;
        _lastTriNum = (int)csTriangles.size();
        if (!csTriangles.empty()) ApplyMeshes(csTriangles, offsets);
        _nextUpdateblocks.clear();
        return;
    }
    struct Hash {
        size_t operator()(const Int3 &k) const { return (size_t)(unsigned)k.GetHashCode(); }
    };
    std::unordered_set<Int3, Hash> updateBlocks;
    std::vector<Int3> ordered;  // HashSet order is arbitrary in the reference; first-insertion order here
    const int ez = _height + 2, ey = _elevation + 2;
    while (!_modifierQueue.empty()) {
        std::shared_ptr<TerrainModifier> modifier = _modifierQueue.front();
        _modifierQueue.pop_front();

        Vector3 worldLow = (modifier->LowerBound() - TerrainOrigin) / _voxelScale;
        auto floorToInt = [](float v) { return v <= -2147483648.0f ? std::numeric_limits<int>::min() : (int)std::floor(v); };
        auto ceilToInt = [](float v) { return v >= 2147483648.0f ? std::numeric_limits<int>::max() : (int)std::ceil(v); };
        Int3 low(floorToInt(worldLow.x), floorToInt(worldLow.y), floorToInt(worldLow.z));
        low._x = std::max(low._x, 0);
        low._y = std::max(low._y, 0);
        low._z = std::max(low._z, 0);
        Vector3 worldUp = (modifier->UpperBound() - TerrainOrigin) / _voxelScale;
        Int3 up(ceilToInt(worldUp.x), ceilToInt(worldUp.y), ceilToInt(worldUp.z));
        up._x = std::min(up._x, _width + 1);
        up._y = std::min(up._y, _elevation + 1);
        up._z = std::min(up._z, _height + 1);

        // resample density function (VoxelTerrain.cs:284-305)
        for (int x = low._x; x <= up._x; x++)
            for (int y = low._y; y <= up._y; y++)
                for (int z = low._z; z <= up._z; z++) {
                    Vector3 worldPos = Vector3((float)x, (float)y, (float)z) * _voxelScale + TerrainOrigin;
                    float &s = _voxelSamples[((size_t)x * ey + y) * ez + z];
                    if (modifier->AddOrErode) {
                        float md = Clamp(modifier->QueryDensity(worldPos), voidDensity(), fullDensity());
                        s = std::max(s, md);
                    } else {
                        float minus_md = -Clamp(modifier->QueryDensity(worldPos), voidDensity(), fullDensity());
                        s = Clamp(std::min(s, minus_md), voidDensity(), fullDensity());
                    }
                }

        // dirty blocks: inclusive on both ends (VoxelTerrain.cs:307-317)
        for (int x = 0; x < _width / blockSize; x++)
            for (int y = 0; y < _elevation / blockSize; y++)
                for (int z = 0; z < _height / blockSize; z++)
                    if ((up._x >= x * blockSize && low._x <= x * blockSize + blockSize) &&
                        (up._y >= y * blockSize && low._y <= y * blockSize + blockSize) &&
                        (up._z >= z * blockSize && low._z <= z * blockSize + blockSize)) {
                        Int3 key(x, y, z);
                        if (updateBlocks.insert(key).second) ordered.push_back(key);
                    }
    }
    _nextUpdateblocks = ordered;
    _lastUpdateBlocks = ordered;
    if (!_nextUpdateblocks.empty()) BatchUpdate();
    _nextUpdateblocks.clear();
}

// VoxelTerrain.cs:330-477.  Steps 1-8 of the reference (tile gather, upload, three dispatches, two
// read-backs) collapse into one Extract call; binning by _block (VoxelTerrain.cs:437-446) becomes
// slicing because the library returns triangles grouped by block in list order.
void VoxelTerrain::BatchUpdate()
{
    if (_nextUpdateblocks.empty()) return;
    std::vector<CSTriangle> csTriangles;
    std::vector<int> offsets;
    _backend->Extract(_voxelSamples.data(), _width, _elevation, _height, _nextUpdateblocks, csTriangles, offsets);
    _lastTriNum = (int)csTriangles.size();
    if (csTriangles.empty()) return;  // "no triangles, early exit" keeps the old meshes (VoxelTerrain.cs:396-405)
    ApplyMeshes(csTriangles, offsets);
}

std::vector<float> VoxelTerrain::DeviceSamples() const
{
    std::vector<float> out;
    if (!_deviceResident || !_backend) throw UnityException("DeviceSamples needs an initialised device-resident terrain");
    _backend->TerrainReadSamples(out);
    return out;
}

// VoxelTerrain.cs:430-465: per dirty block, vertices (scaled by _voxelScale), normals, trivial indices
void VoxelTerrain::ApplyMeshes(const std::vector<CSTriangle> &csTriangles, const std::vector<int> &offsets)
{
    const int nby = _elevation / blockSize, nbz = _height / blockSize;
    for (size_t i = 0; i < _nextUpdateblocks.size(); i++) {
        const Int3 &b = _nextUpdateblocks[i];
        BlockMesh &mesh = _blocks[((size_t)b._x * nby + b._y) * nbz + b._z];
        mesh.Clear();  // VoxelTerrain.cs:453
        for (int t = offsets[i]; t < offsets[i + 1]; t++) {
            const CSTriangle &vt = csTriangles[(size_t)t];
            mesh.vertices.push_back(Vector3(vt._position0[0], vt._position0[1], vt._position0[2]) * _voxelScale);
            mesh.vertices.push_back(Vector3(vt._position1[0], vt._position1[1], vt._position1[2]) * _voxelScale);
            mesh.vertices.push_back(Vector3(vt._position2[0], vt._position2[1], vt._position2[2]) * _voxelScale);
            mesh.normals.push_back(Vector3(vt._normal0[0], vt._normal0[1], vt._normal0[2]));
            mesh.normals.push_back(Vector3(vt._normal1[0], vt._normal1[1], vt._normal1[2]));
            mesh.normals.push_back(Vector3(vt._normal2[0], vt._normal2[1], vt._normal2[2]));
        }
        mesh.triangles.resize(mesh.vertices.size());
        for (size_t k = 0; k < mesh.triangles.size(); k++) mesh.triangles[k] = (int)k;  // Enumerable.Range, VoxelTerrain.cs:457
    }
}

}  // namespace Render
}  // namespace PGRTerrain
