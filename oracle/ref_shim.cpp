/*
 * ref_shim.cpp -- extern "C" doorway into the REFERENCE's own kdtree.cpp, which
 * the Makefile compiles unmodified from /root/reference/src (never copied into
 * this repo) into oracle/_ref/libkdtree_ref.so.  Test infrastructure only.
 */
#include "kdtree.hpp" /* the reference's header, found via -I/root/reference/src */
#include <vector>
#include <cstring>

extern "C" int ref_node_size() { return (int)sizeof(KDTree::Node); }

extern "C" void ref_kd_create(const float *pts_xyzw, int n, void *list)
{
    std::vector<glm::vec4> in(n);
    for (int i = 0; i < n; i++) in[i] = glm::vec4(pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]);
    KDTree::Create(in, (KDTree::Node *)list);
}
extern "C" void ref_kd_insert_node(const float *p, void *list, int list_size)
{
    KDTree::InsertNode(glm::vec4(p[0], p[1], p[2], p[3]), (KDTree::Node *)list, list_size);
}
extern "C" void ref_kd_balance(void *list, int n) { KDTree::Balance((KDTree::Node *)list, n); }
