// kdtree.hpp -- KDTree::Node and the host map-structure API of the reference (src/kdtree.hpp:14-33),
// implemented by libpfslam_hip.so's pfslam_kd_create / pfslam_kd_insert_node / pfslam_kd_balance.
#pragma once
#include <vector>
#include "pf_glm.h"

namespace KDTree {
class Node {
public:
    Node() : axis(0), left(-1), right(-1), parent(-1), value(0, 0, 0, 0) {}
    Node(glm::vec4 p, int state, int source) : axis(state), left(-1), right(-1), parent(source), value(p) {}
    int axis, left, right, parent;
    glm::vec4 value;
};
static_assert(sizeof(Node) == 32, "KDTree::Node must stay 32 bytes");
void Create(std::vector<glm::vec4> input, Node *list);
void InsertList(std::vector<glm::vec4> input, Node *list, int idx, int parent);
void InsertNode(glm::vec4 point, Node *list, int listSize);
void Balance(Node *list, int listSize);
} // namespace KDTree
