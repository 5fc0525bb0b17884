// kernel.h -- drop-in for the reference's src/kernel.h:14-24: same free functions, same structs, implemented on
// libpfslam_hip.so (MI355X).  pbo arguments are accepted and ignored exactly as particleFilter() ignores its own
// (kernel.cu:1702, H9); drawMap is a no-op (visualisation is out of scope).
#pragma once
#include <vector>
#include "kdtree.hpp"
#include "lidar.h"
#include "scene.h"

#ifndef __HIP_PLATFORM_AMD__
struct uchar4 { unsigned char x, y, z, w; };
#endif

// The reference chooses between its GPU and CPU branches of the 2-D path at compile time (kernel.h:9-12; no #ifndef there, so
// they cannot be overridden).  Every stage runs on the GPU here; the macros exist so that callers naming them still compile.
#ifndef GPU_MOTION
#define GPU_MOTION 1
#endif
#ifndef GPU_MEASUREMENT
#define GPU_MEASUREMENT 1
#endif
#ifndef GPU_MAP
#define GPU_MAP 1
#endif
#ifndef GPU_RESAMPLE
#define GPU_RESAMPLE 1
#endif

// floor(log2 x) and ceil(log2 x) helpers of kernel.h:26-36 (x >= 1; ilog2(0) = 0 as there)
inline int ilog2(int x) { return x > 1 ? 31 - __builtin_clz((unsigned)x) : 0; }
inline int ilog2ceil(int x) { return ilog2(x - 1) + 1; }

void particleFilterInit(Scene *scene);
void particleFilterFree();
void particleFilter(uchar4 *pbo, int frame, Lidar *lidar);
void drawMap(uchar4 *pbo);
void getPCData(Particle **ptrParticles, MAP_TYPE **ptrMap, KDTree::Node **ptrKD, int *nParticles, int *nKD, glm::vec3 &pos);
void particleFilterInitPC();
void particleFilterFreePC();
// declared by the reference (kernel.h:24) but never defined there; here it is the point-cloud frame: particleFilter without
// the unused pbo
void particleFilterPC(int frame, Lidar *lidar);

// PARTICLE_COUNT is a compile-time 1000 in the reference (kernel.cu:30); here it is a runtime setting read by the
// next particleFilterInit (also: environment variable PFSLAM_PARTICLES).
void pfslamSetParticleCount(int n);
// The reference keeps both map representations in the source and wires the point-cloud one (kernel.cu:1730-1745);
// this selects which stages particleFilter() runs: false = KD point cloud (default), true = 2-D occupancy grid
// (PFMeasurementUpdate / PFUpdateMap).  Also: environment variable PFSLAM_MAP=grid.
void pfslamUseGridMap(bool on);
// The reference has UpdateTopology() / CheckLoopClosure() commented out at the end of particleFilter (kernel.cu:1750-1751):
// this enables them there (also: environment variable PFSLAM_TOPOLOGY=1).  pfslamLoopClosures returns the
// (candidate node, visible node) pairs the last frame proposed.
void pfslamUseTopology(bool on);
std::vector<std::pair<int, int>> pfslamLoopClosures();
// Map export for an end-to-end comparison: the point-cloud map as the reference's viewer filters it
// (nodes with w > -100, main.cpp:269-284) -> PREFIX.kd.bin (float x, y, z, w per point, in node order) + PREFIX.kd.csv, and
// the 2-D occupancy grid -> PREFIX.grid.i8 (dim.x * dim.y signed bytes, cell (x, y) at x * dim.x + y) + PREFIX.grid.pgm
// (value + 128, one row per x).  Returns the number of exported points.
int pfslamExportMap(const char *prefix);

// error convention of the reference: print and exit (kernel.h:42-60).  checkCUDAErrorFn(msg, file, line) waits for the device
// (every frame in flight is booked: a deferred error of the frame pipeline surfaces here) and exits on failure.
void checkPfslamErrorFn(int rc, const char *msg, const char *file, int line);
void checkCUDAErrorFn(const char *msg, const char *file, int line);
#define checkCUDAError(msg) checkCUDAErrorFn(msg, __FILE__, __LINE__)
