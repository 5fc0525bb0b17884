// selftest.cpp -- no-GPU check of the C++ host layer: loaders and the KDTree facade.
//   pfslam_host_selftest <scene.txt> <lidar.f32> <cloud.txt>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include "kernel.h"
#include "pointcloud.h"
#include "../../include/pfslam.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    Scene scene(argv[1]);
    if (scene.maps.size() != 1) return 10;
    {   // derived camera fields, in the reference's order (scene.cpp:107-121)
        const Camera &c = scene.state.camera;
        printf("camera fov %.6f %.6f pixel %.9f %.9f view %.6f %.6f %.6f right_is_nan %d image %zu\n", c.fov.x, c.fov.y, c.pixelLength.x,
               c.pixelLength.y, c.view.x, c.view.y, c.view.z, (int)(c.right.x != c.right.x), scene.state.image.size());
    }
    printf("map %.6f %.6f %.6f cam %d %d eye %.3f %.3f %.3f file %s\n", scene.maps[0].scale.x, scene.maps[0].scale.y,
           scene.maps[0].resolution.x, scene.state.camera.resolution.x, scene.state.camera.resolution.y,
           scene.state.camera.position.x, scene.state.camera.position.y, scene.state.camera.position.z,
           scene.state.imageName.c_str());
    Lidar lidar(argv[2]);
    printf("lidar %zu %zu %.6f %.6f\n", lidar.scans.size(), lidar.scans[0].size(), lidar.scans[0][0], lidar.scans.back().back());
    Pointcloud pc(argv[3]);
    printf("cloud %zu %.0f %.0f %.0f %.0f\n", pc.points.size(), pc.points[1].x, pc.points[1].y, pc.points[1].z, pc.points[1].w);
    // the small helpers and macros of kernel.h:9-36 and the graph types of sceneStructs.h:47-60 exist for source compatibility
    if (ilog2(1) != 0 || ilog2(2) != 1 || ilog2(1000) != 9 || ilog2ceil(1000) != 10 || ilog2ceil(1024) != 10 || ilog2ceil(1025) != 11) return 20;
    if (!(GPU_MOTION && GPU_MEASUREMENT && GPU_MAP && GPU_RESAMPLE)) return 21;
    {
        Cluster c;
        c.id = 0; c.nodeIdx = 0;
        c.nodes.push_back(Node{glm::vec2(0.0f, 0.0f), 0.0f});
        c.edges.push_back(std::vector<unsigned int>());
        if (c.nodes.size() != 1 || sizeof(Node) != 12) return 22;
    }
    checkCUDAError("selftest"); // no handle yet: returns
    // KDTree facade == C-ABI
    std::vector<glm::vec4> pts;
    for (int i = 0; i < 257; i++) pts.push_back(glm::vec4((float)((i * 37) % 19) * 0.025f, (float)((i * 11) % 23) * 0.025f, 0.0f, (float)(i % 7)));
    std::vector<KDTree::Node> a(300);
    std::vector<pfslam_node> b(300);
    KDTree::Create(pts, a.data());
    pfslam_kd_create(reinterpret_cast<const float *>(pts.data()), 257, b.data());
    KDTree::InsertNode(glm::vec4(0.3f, 0.1f, 0.0f, -100.0f), a.data(), 257);
    const float p[4] = {0.3f, 0.1f, 0.0f, -100.0f};
    pfslam_kd_insert_node(p, b.data(), 257);
    if (memcmp(a.data(), b.data(), 258 * 32) != 0) return 11;
    KDTree::Balance(a.data(), 258);
    pfslam_kd_balance(b.data(), 258);
    if (memcmp(a.data(), b.data(), 258 * 32) != 0) return 12;
    // KDTree::Create == sort on x + InsertList(input, list, 0, -1) (kdtree.cpp:25-29); and a sub-tree below an existing node
    {
        std::vector<glm::vec4> all;
        for (int i = 0; i < 258; i++) all.push_back(a[i].value);
        std::vector<KDTree::Node> viaCreate(300), viaList(300);
        KDTree::Create(all, viaCreate.data());
        std::sort(all.begin(), all.end(), [](const glm::vec4 &p, const glm::vec4 &q) { return p.x < q.x; });
        KDTree::InsertList(all, viaList.data(), 0, -1);
        if (memcmp(viaCreate.data(), viaList.data(), 258 * 32) != 0) return 13;
        std::vector<glm::vec4> few;
        for (int i = 0; i < 5; i++) few.push_back(glm::vec4(0.1f * i, 0.2f, 0.0f, 1.0f));
        KDTree::InsertList(few, viaList.data(), 258, 0); // axis = parent's + 1 = y: all keys tie, mid = 2
        if (viaList[258].axis != 1 || viaList[258].parent != 0 || viaList[258].left != 259 || viaList[258].right != 261) return 14;
    }
    printf("kdtree ok root %d %d %d\n", a[0].axis, a[0].left, a[0].right);
    return 0;
}
