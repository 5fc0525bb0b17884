// pointcloud.h -- Pointcloud (src/pointcloud.h:12-19): text file of integer triples, one point per line;
// the reference stores them as vec4(third, first, second, line index) (pointcloud.cpp:22).
#pragma once
#include <string>
#include <vector>
#include "pf_glm.h"

class Pointcloud {
public:
    explicit Pointcloud(std::string filename);
    ~Pointcloud() {}
    std::vector<glm::vec4> points;
};
