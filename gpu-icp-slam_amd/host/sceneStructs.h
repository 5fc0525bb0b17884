// sceneStructs.h -- the reference's shared structs (src/sceneStructs.h:10-60) with identical layouts.
#pragma once
#include <string>
#include <vector>
#include "pf_glm.h"

#define MAP_TYPE char // sceneStructs.h:10

struct Camera {
    glm::ivec2 resolution;
    glm::vec3 position, lookAt, view, up, right;
    glm::vec2 fov, pixelLength;
};
struct RenderState {
    Camera camera;
    std::vector<glm::vec3> image;
    std::string imageName;
};
struct ParticleHistory { std::vector<glm::vec3> patchPos; };
struct Particle { // 32 bytes: pos@0, w@12, cluster@16, map@24
    glm::vec3 pos;
    float w;
    unsigned char cluster;
    ParticleHistory *map;
};
struct Patch { // 40 bytes
    glm::vec3 scale;
    glm::vec3 resolution;
    MAP_TYPE *grid;
    unsigned char uid;
};
// topology graph of a map cluster (sceneStructs.h:47-60); the library keeps cluster 0's graph inside the handle
// (pfslam_get_topology), these types are here for source compatibility with callers that name them
struct Node { // graph node: position and path length to the current node
    glm::vec2 pos;
    float dist;
};
struct Cluster {
    unsigned char id;
    unsigned int nodeIdx;
    std::vector<unsigned int> patchList;
    std::vector<Node> nodes;
    std::vector<std::vector<unsigned int>> edges;
};
static_assert(sizeof(Node) == 12, "graph node layout");
static_assert(sizeof(Particle) == 32, "Particle must stay 32 bytes (getPCData hands out raw pointers)");
static_assert(sizeof(Patch) == 40, "Patch layout");
