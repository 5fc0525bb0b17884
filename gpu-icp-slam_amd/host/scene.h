// scene.h -- Scene (src/scene.h:13-24): parses the reference's text config (CAMERA / MAP blocks,
// data/map_settings.txt) into maps[0].{scale,resolution} and state.camera.
#pragma once
#include <iosfwd>
#include <string>
#include <vector>
#include "sceneStructs.h"

class Scene {
public:
    // same public surface as the reference class: ctor from a config path, maps, state
    explicit Scene(std::string filename);
    ~Scene() {}
    std::vector<Patch> maps;
    RenderState state;

private:
    // one block of "KEY value..." lines up to the next blank line
    void parse_map_block(std::istream &in);
    void parse_camera_block(std::istream &in);
};
