// lidar.h -- Lidar (src/lidar.h:13-18): scans[frame] = 1081 ranges.  `filename` ending in .mat is read as the
// reference's own format (MATLAB Level-5 file, cell array `lidar` of structs with a `scan` field, lidar.cpp:17-49)
// by the built-in reader (mat5_reader.cpp: no libmat needed); anything else as a flat little-endian float32 file of
// frames x 1081 ranges.  Scans can also be handed over from memory.
#pragma once
#include <string>
#include <vector>

class Lidar {
public:
    explicit Lidar(std::string filename);
    explicit Lidar(std::vector<std::vector<float>> in) : scans(std::move(in)) {}
    ~Lidar() {}
    std::vector<std::vector<float>> scans;
};
