/*
 * vtmc.h -- C ABI of the MI355X-native marching-cubes extractor (libvtmc.so).
 *
 * Drop-in boundary for the GPU section of PGRTerrain.Render.VoxelTerrain.BatchUpdate
 * (reference: Unity-Project/Assets/Scripts/VoxelTerrain.cs:330-477), which today drives three
 * Unity compute shaders (Shaders/SampleNormal.compute, CollectTriNum.compute, MarchingCube.compute)
 * through nine ComputeBuffer bindings (VoxelTerrain.cs:370-421).  Every entry point is cdecl,
 * `extern "C"`, takes only plain pointers / integers / the 76-byte POD below, and never throws:
 * the return value is a status (0 = OK, negative = error, text via vtmc_last_error).
 *
 * Ownership: host memory passed in is borrowed for the duration of the call only; all device
 * memory is owned by the context.  Calls on one context are serialised by the caller
 * (the reference calls from the Unity main thread only, TerrainEngine.cs:145-149); different
 * contexts may be used from different threads.  Host entry points block until results are complete.
 *
 * Canonical triangle order (the reference's order is whatever its atomic append produces,
 * MarchingCube.compute:160-162): (block index in the submitted list, cell x + 8y + 64z, table
 * triangle i).  The host can therefore slice per block with block_tri_offsets instead of binning
 * (VoxelTerrain.cs:437-446).
 */
#ifndef VTMC_H
#define VTMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTMC_BLOCK_SIZE 8          /* VoxelTerrain.cs:54  blockSize */
#define VTMC_TILE_SAMPLES 1000     /* (blockSize+2)^3, VoxelTerrain.cs:337-341 */
#define VTMC_MAX_TRIS_PER_CELL 5   /* VoxelTerrain.cs:480 maxTriNumPerCell */

/* status codes */
#define VTMC_OK 0
#define VTMC_ERR_INVALID_ARG (-1)  /* null pointer, negative count ... */
#define VTMC_ERR_DIMS (-2)         /* dims not a multiple of 8 (VoxelTerrain.cs:138-139) or block out of range */
#define VTMC_ERR_CAPACITY (-3)     /* destination smaller than the triangle count */
#define VTMC_ERR_DEVICE (-4)       /* HIP runtime error (text in vtmc_last_error) */
#define VTMC_ERR_NO_RESULT (-5)    /* read_* before any extract_* */
#define VTMC_ERR_TOO_LARGE (-6)    /* more than 2^31-1 triangles or blocks */

/* Wire format of one triangle -- replaces the private CSTriangle read-back struct
 * (VoxelTerrain.cs:23-37; HLSL twin MarchingCube.compute:18-27).  76 bytes, packed.  Positions
 * are block-local in cell units [0,8]; `block` is the index into the submitted block list. */
typedef struct vtmc_triangle {
    float position0[3];
    float position1[3];
    float position2[3];
    float normal0[3];
    float normal1[3];
    float normal2[3];
    int32_t block;
} vtmc_triangle;

typedef struct vtmc_ctx vtmc_ctx;

/* Replaces the table uploads of VoxelTerrain.Init (VoxelTerrain.cs:151-156): binds HIP device
 * `device`, uploads the lookup tables, creates the stream and reusable scratch. */
int32_t vtmc_create(int32_t device, vtmc_ctx **out_ctx);

/* Replaces the ComputeBuffer.Release calls of VoxelTerrain.Free (VoxelTerrain.cs:228-244) and of
 * BatchUpdate's epilogue (VoxelTerrain.cs:469-476). */
int32_t vtmc_destroy(vtmc_ctx *ctx);

/* UTF-8 text of the last error on this context ("" if none).  ctx may be NULL (global create error). */
const char *vtmc_last_error(const vtmc_ctx *ctx);

/* Replaces bufferSamples.SetData + the three Dispatch calls + bufferTriNum.GetData
 * (VoxelTerrain.cs:365-395).  `samples` = n_blocks tiles of 10x10x10 floats, x fastest, exactly the
 * array BatchUpdate builds (VoxelTerrain.cs:341-361).  *tri_count receives T (0 is success, the
 * reference's early-out VoxelTerrain.cs:396-405). */
int32_t vtmc_extract_blocks(vtmc_ctx *ctx, const float *samples, int32_t n_blocks, int32_t *tri_count);

/* New: removes the tile gather (VoxelTerrain.cs:337-361).  Reads the (nx+2, ny+2, nz+2)-sample
 * grid in place: sample (x,y,z) = grid[x*stride_x + y*stride_y + z*stride_z] (element strides), so
 * a pinned C# float[W+2,E+2,H+2] (z fastest, VoxelTerrain.cs:145) is passed with strides
 * ((E+2)*(H+2), H+2, 1).  block_list = n_blocks (bx,by,bz) triples (the dirty list
 * VoxelTerrain.cs:321); NULL = every block, ordered bx + nbx*(by + nby*bz), n_blocks ignored. */
int32_t vtmc_extract_grid(vtmc_ctx *ctx, const float *grid, int32_t nx, int32_t ny, int32_t nz,
                          int64_t stride_x, int64_t stride_y, int64_t stride_z,
                          const int32_t *block_list, int32_t n_blocks, int32_t *tri_count);

/* Multi-GPU host entry: the grid is cut into chunks of chunk_cells^3 cells, chunk c
 * (c = cx + ncx*(cy + ncy*cz)) belongs to rank c % world_size.  Extracts this rank's chunks only,
 * chunk-major, blocks in canonical order inside each chunk.  chunk_counts receives for each LOCAL
 * chunk {vertex count, triangle count}; *n_local_chunks how many.  No collective is issued here:
 * vtmc_allgather_volume_counts (below) is the RCCL all-gather that follows it. */
int32_t vtmc_extract_grid_sharded(vtmc_ctx *ctx, const float *grid, int32_t nx, int32_t ny, int32_t nz,
                                  int64_t stride_x, int64_t stride_y, int64_t stride_z,
                                  int32_t chunk_cells, int32_t rank, int32_t world_size,
                                  uint32_t *chunk_counts, int32_t chunk_counts_capacity,
                                  int32_t *n_local_chunks, int32_t *tri_count);

/* Replaces bufferMeshes.GetData(csTriangles) (VoxelTerrain.cs:426-427).  Copies the T triangles of
 * the last extract_* in canonical order.  block_tri_offsets (optional, n_blocks+1 entries) receives
 * the exclusive prefix of per-block triangle counts. */
int32_t vtmc_read_triangles(vtmc_ctx *ctx, vtmc_triangle *dst, int64_t capacity, int32_t *block_tri_offsets);

/* The _CornerFlags buffer of the last extract_* (CollectTriNum.compute:56-62) as one byte per
 * cell: dst[512*b + x + 8y + 64z].  Parity / debugging aid; capacity in bytes. */
int32_t vtmc_read_cases(vtmc_ctx *ctx, uint8_t *dst, int64_t capacity);

/* Number of blocks / triangles of the last extract_*. */
int32_t vtmc_last_counts(const vtmc_ctx *ctx, int32_t *n_blocks, int32_t *tri_count);

/* ------------------------------------------------------------------------------------------
 * Device-resident entry points: same extraction, inputs already in HBM (what a GPU-resident host,
 * the sharded driver and bench.py use).  Pointers prefixed d_ are device pointers on ctx's device.
 * ------------------------------------------------------------------------------------------ */

/* A batch of equally shaped volumes: volume v starts at d_samples + v*volume_stride and holds
 * (nx+2, ny+2, nz+2) samples addressed with the element strides.  One 1026^3 grid is a batch of 1;
 * the reference's tile buffer is a batch of n_blocks volumes with nx=ny=nz=8, strides (1,10,100),
 * volume_stride 1000; "1024^3 as 8^3 chunks of 128^3" is a batch of 512 with nx=ny=nz=128. */
typedef struct vtmc_volume_batch {
    const float *d_samples;
    int32_t nx, ny, nz;
    int64_t stride_x, stride_y, stride_z;
    int32_t n_volumes;
    int64_t volume_stride;
} vtmc_volume_batch;

#define VTMC_FLAG_WANT_CASES 1u    /* also materialise the per-cell case bytes (slower generic classify) */
#define VTMC_FLAG_NO_DENSE_PATH 2u /* force the per-block classify kernel (A/B testing) */

/* Extract every block of every volume (block id = v*blocks_per_volume + bx + nbx*(by + nby*bz)).
 * `stream` is a hipStream_t (NULL = the context's own stream).  Blocks until T is known. */
int32_t vtmc_extract_volumes_device(vtmc_ctx *ctx, const vtmc_volume_batch *batch, void *stream,
                                    uint32_t flags, int64_t *tri_count);

/* The same extract in two halves, for hosts that keep the CPU out of the step (the multi-GPU driver):
 * _async queues classify -> scan -> emit on `stream` and returns at once -- {T, nActive} stay in
 * device memory, there is no mid-pipeline read-back (VoxelTerrain.cs:394-395) --; after it the per-
 * volume counts are final on the stream, so vtmc_copy_volume_counts_device / vtmc_allgather_volume_counts
 * may be queued behind it.  vtmc_extract_finish waits for THIS extract (an event behind its emit
 * launch -- not for work the caller queued behind it on the stream: the next batch's sampler, a
 * collective, copies; a caller that wants those too synchronises its stream), returns T and -- when
 * the triangle buffer turned out too small and the emit kernel refused to run -- grows it and runs
 * the emit stage again.  Exactly one _finish per _async; no other extract_* of this context in between. */
int32_t vtmc_extract_volumes_device_async(vtmc_ctx *ctx, const vtmc_volume_batch *batch, void *stream, uint32_t flags);
int32_t vtmc_extract_finish(vtmc_ctx *ctx, int64_t *tri_count);

/* A stream of the context (a hipStream_t).  own_queue = 0: the context's own stream, what `stream` = NULL means everywhere above.
 * own_queue = 1: a second stream that sits on a HARDWARE QUEUE OF ITS OWN (taken on first request): ordinary HIP streams share a handful
 * of hardware queues, and two contexts whose streams land on one queue run their steps strictly in turn; a host that keeps several steps
 * in flight passes each context's own-queue stream to vtmc_extract_volumes_device_async and the steps overlap where one kernel drains and
 * the next ramps up (bench.py --streams 2).
 * LIFETIME: the context stops using the stream at vtmc_destroy (which drains it), but the HANDLE stays valid until the process exits: the
 * library never destroys a stream, it parks it and gives it to the next context created on that device.  Host-side objects that still refer
 * to it after vtmc_destroy -- events recorded on it, a framework's stream wrapper, a caching allocator that records on it when it frees a
 * pinned buffer -- are therefore safe, whatever order they are released in.  Work the HOST queues on the handle after vtmc_destroy simply
 * shares the stream with that next context.  See INTEGRATION.md ("Streams").
 * (The reference has one queue: every Dispatch / GetData of BatchUpdate is in program order on Unity's graphics device,
 * VoxelTerrain.cs:365-427.) */
int32_t vtmc_context_stream(vtmc_ctx *ctx, int32_t own_queue, void **stream);

/* Destroys the streams the library holds for contexts that no longer exist (the parked ones; a live context keeps its own).  Optional: a host
 * calls it at a point where nothing of its own refers to a handle of a destroyed context any more -- typically once, before it exits.  Needed
 * in one situation only: under a profiler (rocprofv3) a process that ends with streams on hardware queues of their own still alive crashes in
 * the profiler's finalisation and loses the profile; bench.py and the tools that are run under rocprofv3 call it for that reason.  Returns the
 * number of streams destroyed. */
int32_t vtmc_release_streams(void);

/* Device pointers to the results of the last extract_* (valid until the next extract_* / destroy):
 * triangles (T x 76 B), block_tri_offsets (n_blocks+1 x u32), volume_counts (n_volumes x
 * {vertices, triangles} u32 -- the array SURVEY.md 8e all-gathers).  Any out pointer may be NULL. */
int32_t vtmc_device_results(vtmc_ctx *ctx, const vtmc_triangle **d_triangles,
                            const uint32_t **d_block_tri_offsets, const uint32_t **d_volume_counts);

/* Copies volume_counts of the last extract_* (n_volumes x {vertices, triangles} u32) into a
 * caller-owned DEVICE buffer on `stream` (NULL = the context's stream), asynchronously: the buffer a
 * multi-GPU caller hands to its all-gather (RCCL), SURVEY.md 8e. */
int32_t vtmc_copy_volume_counts_device(vtmc_ctx *ctx, uint32_t *d_dst, int32_t capacity_volumes, void *stream);

/* Pre-size the triangle buffer (otherwise it grows on demand and the emit stage is re-run once). */
int32_t vtmc_reserve_triangles(vtmc_ctx *ctx, int64_t capacity);

/* Per-stage device time of the last extract_* in milliseconds, measured with HIP events on the
 * stream the kernels ran on: ms[0] classify+count, ms[1] scan,
 * ms[2] emit, ms[3] whole call.
 * The reference's only timing hook is the commented-out timer at VoxelTerrain.cs:363/467. */
int32_t vtmc_last_stage_ms(vtmc_ctx *ctx, float ms[4]);

/* Selects a kernel variant / launch shape, mainly for A/B measurements in one process.  Every key and value the library accepts keeps
 * results within the parity bar and is compared with the CPU oracle by tests/test_tuning_matrix.py; an unknown key or a value outside
 * a key's range answers VTMC_ERR_INVALID_ARG and changes nothing.  Keys: "emit_fast_math" (1: v_rcp / v_rsq / fma, default; 0: correctly
 * rounded, bit-compatible with the CPU oracle), "emit_once" (default 1: with emit_fast_math, every welded vertex of a block is evaluated
 * once and the 76-byte records are expanded from LDS; 0: per triangle corner), "emit_dynamic" (default 1: per-XCD ticket counters; 0: a
 * static round-robin over the list of non-empty blocks), "emit_sub_log2" (0-4, default 1: 2^s ticket counters per XCD),
 * "emit_row_masks" (default 1: only tile rows next to cells with triangles are fetched), "emit_wgs_per_cu" (0-8; 0, default: the
 * kernel's own residency),
 * "classify_wgs_per_cu" (0 or 2-7, default 3: residency cap of the streaming classify kernel; 0: none), "density_wgs_per_cu" (0, 2 or 3:
 * residency cap of the synthetic sampler), "stage_events" (default 1: HIP events between the three kernels for vtmc_last_stage_ms; 0: only around
 * the step), "gather_beside" (default 0: the all-gather of a queued extract runs behind the emit kernel on the caller's stream; 1: beside
 * it on the context's second stream), "place_outputs" (0-16, default 0: see vtmc_last_placement below).  The diagnostic keys "emit_ablate" / "classify_ablate" / "density_ablate" (parts of a kernel
 * switched off, output INVALID) exist only in -DVTMC_DIAGNOSTICS builds of the library (tools/build_diagnostics.py).
 * "fill_keeps_signs" (default 0) is a contract, not a variant: with 1, vtmc_density_fill_device[_async] also leaves
 * one sign bit per sample in context memory, and an extract by the SAME context of exactly that buffer (pointer,
 * dims, strides, volume count) classifies from those bits instead of the samples (1/32 of the bytes; results are
 * identical).  The caller vouches that nothing wrote to the buffer between the fill and the extract; any other
 * fill by the context, setting the key again, or "invalidate_signs" (any value) drops the bits -- a caller whose allocator may hand the
 * same address to a new buffer of the same shape calls that when it frees the old one.  streaming.ChunkStream sets it. */
int32_t vtmc_set_tuning(vtmc_ctx *ctx, const char *key, int32_t value);

/* OUTPUT PLACEMENT (tuning key "place_outputs" = K, 2-16; default 0 = off).  The emit kernel's time is a property of the pair (allocation of the
 * input field, allocation of the output buffers): the identical kernel on the identical input runs 0.86 ... 1.00 ms by which allocation it
 * writes (profiles/r06/placement_probe.txt).  With K > 1, whenever the library has just (re)allocated its output buffers -- a context's first
 * extract, a growth -- it runs the emit stage of the extract at hand into K - 1 further allocations of the same size, times each and keeps the
 * fastest (an autotuner's move; the result is complete and identical in every candidate).  Cost: 2 (K - 1) emit launches and K allocations of the
 * output held at once while the trial runs (they must differ: one freed and made again gets its old pages back), once per (re)allocation.  vtmc_last_placement reports the last trial: the emit stage's milliseconds per candidate (ms[0] = the
 * allocation that was there), how many were tried (0: no trial yet) and which one was kept. */
int32_t vtmc_last_placement(const vtmc_ctx *ctx, float ms[16], int32_t *n_candidates, int32_t *kept);

/* Synthetic density sampler (the reference has no noise field of its own):
 * density = sum_{o<octaves} gain^o * perlin(p*frequency*lacunarity^o) - (p.y - ramp_center)*ramp_scale,
 * p = volume origin + sample index, FP32, Perlin 2002 improved noise with a SplitMix64 permutation. */
typedef struct vtmc_density_params {
    uint64_t seed;
    float frequency;
    int32_t octaves;
    float lacunarity;
    float gain;
    float ramp_scale;
    float ramp_center;
} vtmc_density_params;

/* Fill n_volumes volumes of (dim_x, dim_y, dim_z) samples; volume v has global sample origin
 * origins[3v..3v+2] (host array) and is written at d_out + v*volume_stride with element strides. */
int32_t vtmc_density_fill_device(vtmc_ctx *ctx, const vtmc_density_params *params,
                                 const int32_t *origins, int32_t n_volumes,
                                 int32_t dim_x, int32_t dim_y, int32_t dim_z,
                                 int64_t stride_x, int64_t stride_y, int64_t stride_z,
                                 int64_t volume_stride, float *d_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Indexed (welded) output -- new; the reference welds on the CPU afterwards with Mesh.Optimize()
 * (VoxelTerrain.cs:460).  Per block: one vertex per lattice edge with a sign change (all cells
 * around the edge share it).  Every such edge is a cube edge of exactly one OWNER cell -- the cell
 * whose corner 0 is the edge's low point (cube edges 0, 3, 8 of MarchingCube.compute:40-43) or, on the
 * block's x = 8 / y = 8 / z = 8 faces, the boundary cell next to it -- and vertices are ordered by owner
 * cell x + 8y + 64z, then cube edge id; three block-local int32 indices per triangle in the canonical
 * triangle order.  ~27 bytes per triangle instead of 76.  De-indexing reproduces the 76-byte records
 * within the 1e-5 bar (the reference evaluates an edge from either end depending on the cell; here
 * always from its low end).  Faster than the soup on the benchmark field.
 * ------------------------------------------------------------------------------------------ */
typedef struct vtmc_vertex {
    float position[3]; /* block-local, cell units [0,8] */
    float normal[3];
} vtmc_vertex;

#define VTMC_OUTPUT_SOUP 0    /* 76-byte CSTriangle records (default) */
#define VTMC_OUTPUT_INDEXED 1 /* vtmc_vertex + index buffers */

/* Selects what the following extract_* / terrain_update calls produce. */
int32_t vtmc_set_output_mode(vtmc_ctx *ctx, int32_t mode);

/* Vertex count of the last extract in indexed mode (tri count: vtmc_last_counts). */
int32_t vtmc_last_vertex_count(const vtmc_ctx *ctx, int32_t *vertex_count);

/* Copies the indexed mesh of the last extract: V vertices, 3*T indices (block-local: add nothing,
 * a block's vertices are vertices[block_vertex_offsets[b] .. block_vertex_offsets[b+1])), and the
 * two per-block exclusive prefixes (n_blocks+1 entries each, optional). */
int32_t vtmc_read_indexed_mesh(vtmc_ctx *ctx, vtmc_vertex *vertices, int64_t vertex_capacity, int32_t *indices,
                               int64_t tri_capacity, int32_t *block_vertex_offsets, int32_t *block_tri_offsets);

/* Device pointers of the same (valid until the next extract_* / destroy). */
int32_t vtmc_device_indexed_results(vtmc_ctx *ctx, const vtmc_vertex **d_vertices, const int32_t **d_indices,
                                    const uint32_t **d_block_vertex_offsets, const uint32_t **d_block_tri_offsets);

/* ------------------------------------------------------------------------------------------
 * Device-resident terrain: the density grid of VoxelTerrain and its Update() on the GPU.
 * Replaces _voxelSamples (VoxelTerrain.cs:42,145), the per-sample CSG loop of Update
 * (VoxelTerrain.cs:284-305), the dirty-block selection (VoxelTerrain.cs:307-317) and the hand-off to
 * BatchUpdate (VoxelTerrain.cs:321-324) without the grid ever crossing PCIe.
 * ------------------------------------------------------------------------------------------ */

#define VTMC_MOD_PLANE 0     /* TerrainModifier.cs:38-65   f = _height - y            p[0] = _height */
#define VTMC_MOD_SPHERE 1    /* TerrainModifier.cs:70-91   f = _radius - |pos - c|    p[0..2] = _center, p[3] = _radius */
#define VTMC_MOD_CYLINDER 2  /* TerrainModifier.cs:96-152  p[0..2] = _axisStart, p[3..5] = _axisDir (normalised),
                                                           p[6] = _axisLength, p[7] = _radius */
#define VTMC_MOD_HEIGHTMAP 3 /* IslandModifier.cs:34-73    f = bilinear(_heightmap)(x, z) - y; p[0] = _island.width,
                                                           p[1] = _island.height; data = _heightmap (host pointer,
                                                           float[data_dims[0], data_dims[1]] row-major as the C# float[,]) */

/* One queued TerrainModifier (TerrainModifier.cs:19-33).  lower / upper are the values the C#
 * LowerBound / UpperBound properties return (world space): the shim copies them, so Unity's
 * Vector3.ProjectOnPlane stays on the C# side. */
typedef struct vtmc_modifier {
    int32_t kind;
    int32_t add_or_erode; /* 1: add (union, max), 0: erode (difference, clamped min of the negation) */
    float lower[3];
    float upper[3];
    float p[8];
    const float *data;    /* VTMC_MOD_HEIGHTMAP only: borrowed for the call, copied to the device */
    int32_t data_dims[2];
} vtmc_modifier;

/* Replaces the grid allocation + fill of VoxelTerrain.Init (VoxelTerrain.cs:121-149): a
 * (width+2, elevation+2, height+2)-sample grid in HBM, every sample a "void" value in [-2,-1].
 * The reference draws voidDensity / fullDensity from UnityEngine.Random on every read
 * (VoxelTerrain.cs:50-51); here they are a counter-based hash of (seed, event, sample, draw) with
 * the same ranges and the same number of draws, so results are reproducible.  Errors as the
 * reference's: dims not a multiple of 8, more than 1025 samples per axis (VoxelTerrain.cs:138-142). */
int32_t vtmc_terrain_init(vtmc_ctx *ctx, int32_t width, int32_t elevation, int32_t height, float voxel_scale,
                          const float terrain_origin[3], uint64_t seed);

/* Replaces VoxelTerrain.Update (VoxelTerrain.cs:262-325) for a queue of n_mods modifiers, applied
 * in order: AABB -> sample range (floor / ceil, clamped to [0, dim+1]), per-sample density write,
 * union of dirty blocks (a block is dirty on an axis when up >= 8b && low <= 8b+8), then
 * BatchUpdate on that set.  The dirty list is ordered by bx + nbx*(by + nby*bz) (the reference's
 * HashSet order is arbitrary); `block` of a triangle indexes it.  *n_dirty_blocks and *tri_count may
 * be NULL.  Triangles are read with vtmc_read_triangles / vtmc_device_results as after any extract. */
int32_t vtmc_terrain_update(vtmc_ctx *ctx, const vtmc_modifier *mods, int32_t n_mods, int32_t *n_dirty_blocks,
                            int32_t *tri_count);

/* The dirty list of the last vtmc_terrain_update: n (bx,by,bz) triples (_nextUpdateblocks,
 * VoxelTerrain.cs:321). */
int32_t vtmc_terrain_dirty_blocks(vtmc_ctx *ctx, int32_t *dst, int32_t capacity_blocks, int32_t *n_blocks);

/* Copies the density grid to the host: dst[x*stride_x + y*stride_y + z*stride_z] (element strides;
 * a C# float[W+2,E+2,H+2] is ((E+2)*(H+2), H+2, 1)).  Parity / debugging / persistence. */
int32_t vtmc_terrain_read_samples(vtmc_ctx *ctx, float *dst, int64_t stride_x, int64_t stride_y, int64_t stride_z);

/* Device pointer + element strides of the grid (x fastest), for GPU-resident callers. */
int32_t vtmc_terrain_device_grid(vtmc_ctx *ctx, const float **d_samples, int64_t strides[3], int32_t dims[3]);

/* The same fill without the final synchronisation: queued on `stream` (NULL = the context's stream)
 * and ordered only by it, so a streaming driver can generate batch k+1 on one context / stream while
 * batch k is extracted on another (BASELINE config "2048^3 streaming grid, double-buffered chunks").
 * The origins are staged in context-owned device memory: issue fills of ONE context on one stream. */
int32_t vtmc_density_fill_device_async(vtmc_ctx *ctx, const vtmc_density_params *params,
                                       const int32_t *origins, int32_t n_volumes,
                                       int32_t dim_x, int32_t dim_y, int32_t dim_z,
                                       int64_t stride_x, int64_t stride_y, int64_t stride_z,
                                       int64_t volume_stride, float *d_out, void *stream);

/* Device time in milliseconds of the density kernel the last vtmc_density_fill_device[_async] queued
 * (HIP events on the stream it ran on; waits for that kernel only). */
int32_t vtmc_last_fill_ms(vtmc_ctx *ctx, float *ms);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: one context per GPU / process, chunk c -> rank c % world_size, and ONE collective --
 * an RCCL all-gather (over xGMI) of the per-chunk {vertices, triangles} pairs -- after which every
 * rank derives the global offsets with a local exclusive scan.  New in the build: the
 * reference is single-process / single-GPU; the call sits where BatchUpdate hands its results to the
 * host (VoxelTerrain.cs:426-446).  librccl is bound at run time on the first vtmc_comm_* call.
 * ------------------------------------------------------------------------------------------ */
#define VTMC_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

/* Rank 0 draws the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by the host's own
 * means (the C# host: its launcher's socket / file; bench.py: torch.distributed broadcast). */
int32_t vtmc_comm_unique_id(uint8_t id[VTMC_COMM_ID_BYTES]);

/* Collective over all ranks: creates this context's communicator on its device (ncclCommInitRank). */
int32_t vtmc_comm_init_rank(vtmc_ctx *ctx, const uint8_t id[VTMC_COMM_ID_BYTES], int32_t rank, int32_t world_size);
int32_t vtmc_comm_destroy(vtmc_ctx *ctx);

/* `ctx` uses the communicator of `owner` (same device, same process) for vtmc_allgather_volume_counts from now on: contexts that take
 * turns (a step in flight while the host takes the previous one) issue all their collectives through ONE communicator.  ORDER: the
 * collectives of a communicator -- its owner's and every borrower's -- form one chain kept by the owner: a collective queued on another
 * stream than the one before it first waits, on the device, for the previous one's event, so RCCL sees them one after the other, in host
 * call order, exactly as on one stream -- and that order must be the same on every rank.  ONE HOST THREAD drives all contexts that share a
 * communicator (the chain's state on the owner is updated by the borrowers without a lock); contexts with communicators of their own may
 * live on different threads.  `ctx` never destroys the communicator; `owner` must stay alive, and keep it, as long as `ctx` uses it
 * (vtmc_comm_destroy(ctx) or another vtmc_comm_* call on ctx ends the sharing; an owner that goes first detaches its borrowers). */
int32_t vtmc_comm_share(vtmc_ctx *ctx, vtmc_ctx *owner);

/* All-gather of volume_counts of the last extract_* on `stream` (NULL = the context's stream),
 * asynchronously: d_all_counts (device, world_size x volumes_per_rank x {vertices, triangles} u32)
 * receives rank r's pairs at [r * volumes_per_rank ...), zero-padded where a rank owns fewer volumes.
 * No host synchronisation: the caller orders later work on the same stream.
 * Queued behind vtmc_extract_volumes_device_async (before vtmc_extract_finish) whose chunks are whole
 * scan tiles (a multiple of 2048 blocks, e.g. 128^3 cells), the counts have already left the scan
 * kernel: the collective then runs on the context's second stream BESIDE the emit kernel (launched a
 * workgroup per XCD short for it) and `stream` merely waits for its end -- only with the tuning key "gather_beside" = 1
 * (default 0).  Otherwise, and by default, it runs on `stream`, behind the emit kernel. */
int32_t vtmc_allgather_volume_counts(vtmc_ctx *ctx, uint32_t *d_all_counts, int32_t volumes_per_rank, void *stream);

/* Blocking device -> host copy on `stream` (NULL = the context's stream) through the library's own
 * HIP runtime: for hosts that hold device pointers from vtmc_device_results and no HIP binding. */
int32_t vtmc_copy_to_host(vtmc_ctx *ctx, const void *d_src, void *dst, int64_t bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Persisted chunks (layout: volumetricterrain_amd/chunkfile.py -- 64-byte
 * header, then 16-byte aligned sections).  New in the build: the reference keeps its grid in memory
 * only (VoxelTerrain.cs:145-149).  The file image is assembled on the device.
 * ------------------------------------------------------------------------------------------ */
#define VTMC_CHUNK_SAMPLES 1u /* f32[(cells+2)^3], x fastest */
#define VTMC_CHUNK_SOUP 2u    /* 76-byte records, `block` relative to the chunk */
#define VTMC_CHUNK_INDEXED 4u /* vert_offsets, vertices, indices */

/* Writes volume `volume` of the last vtmc_extract_volumes_device / vtmc_extract_grid (its samples when
 * with_samples != 0 -- the input of that extract must still be resident --, its block offsets rebased to
 * 0 and its mesh in the output mode that extract ran in).  origin = global sample index of the chunk. */
int32_t vtmc_chunk_write(vtmc_ctx *ctx, const char *path, int32_t volume, const int32_t origin[3], int32_t with_samples);

typedef struct vtmc_chunk_view {
    int32_t origin[3];
    int32_t cells[3];
    uint32_t flags, n_blocks, n_triangles, n_vertices;
    const float *d_samples;            /* NULL when the section is absent; strides (1, cells[0]+2, (cells[0]+2)*(cells[1]+2)) */
    const uint32_t *d_tri_offsets;     /* n_blocks + 1 */
    const vtmc_triangle *d_triangles;
    const uint32_t *d_vert_offsets;
    const vtmc_vertex *d_vertices;
    const int32_t *d_indices;
} vtmc_chunk_view;

/* Uploads a chunk file as one image and returns DEVICE pointers to its sections (owned by the
 * context, valid until the next vtmc_chunk_read / vtmc_chunk_write / vtmc_destroy): d_samples can be
 * handed straight back to vtmc_extract_volumes_device. */
int32_t vtmc_chunk_read(vtmc_ctx *ctx, const char *path, vtmc_chunk_view *out);

/* Library / build identification: "vtmc <version> gfx950". */
const char *vtmc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VTMC_H */
