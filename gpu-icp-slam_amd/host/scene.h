// scene.h -- Scene (src/scene.h:13-24): parses the reference's text config (CAMERA / MAP blocks,
// data/map_settings.txt) into maps[0].{scale,resolution} and state.camera.
#pragma once
#include <fstream>
#include <string>
#include <vector>
#include "sceneStructs.h"

class Scene {
private:
    std::ifstream fp_in;
    int loadGeom();
    int loadCamera();
public:
    explicit Scene(std::string filename);
    ~Scene() {}
    std::vector<Patch> maps;
    RenderState state;
};
