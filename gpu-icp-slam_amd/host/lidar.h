// lidar.h -- Lidar (src/lidar.h:13-18): scans[frame] = 1081 ranges.  The reference reads a MATLAB .mat through
// libmat (not available); this class reads a flat little-endian float32 file of frames x 1081 ranges
// (tools/mat2bin.py converts train_lidar*.mat), or takes the scans from memory.
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
