// mat5_reader.h -- minimal reader for MATLAB Level-5 MAT-files, just enough for the reference's lidar logs:
// variable `lidar` = cell array, each cell a struct with a numeric field `scan` (src/lidar.cpp:17-49 reads it through
// MathWorks' libmat, which is not redistributable).  Handles little-endian v5 files with or without zlib-compressed
// variables (miCOMPRESSED), small data elements, and numeric storage narrower than the array class.
#pragma once
#include <string>
#include <vector>

// Returns true and fills `scans` (one vector per cell that has a `scan` field, in cell order) on success;
// on failure returns false and sets `err`.
bool mat5_read_lidar_scans(const std::string &path, std::vector<std::vector<float>> &scans, std::string &err,
                           const char *var_name = "lidar", const char *field_name = "scan");
