// pt_layout.h — records shared by the host-side flattener and the gfx950 kernels.
//
// Everything a ray touches in HBM is a 16-byte-aligned record read with
// dwordx4 loads: one lane = one ray, and rays in a wave diverge, so what counts
// is bytes per gather and alignment, not cross-lane contiguity.
#ifndef PT_LAYOUT_H
#define PT_LAYOUT_H

#include <stdint.h>

namespace ptl {

// Kinds of a child slot of a two-child device node.
enum : uint32_t {
  KIND_INTERNAL = 0,  // an internal node of the reference tree: its own box test decides (object.cpp:277)
  KIND_GROUP = 1,     // grouping node that does not exist in the reference (several meshes of one
                      // top-level leaf, or the super-root): the reference visits everything below it
  KIND_LEAF = 2,      // triangle leaf: the reference never box-tests a leaf (object.cpp:266-275)
  KIND_SPLIT = 3      // grouping node inside one reference leaf that was split into smaller leaves; walked like
                      // KIND_GROUP, told apart only where the reference's leaf boundary matters (HitData::hits)
};

// Encoding of a traversal-stack entry / child reference.
//   bit 31 = 0: index of a device node
//   bit 31 = 1: leaf, bits 28..30 = count-1 (1..8 triangles), bits 0..27 = first triangle slot
static const uint32_t REF_LEAF = 0x80000000u;
static const uint32_t REF_FIRST_MASK = 0x0fffffffu;
static const uint32_t MAX_TRI_SLOTS = (1u << 28) - 8u;

// 64-byte node holding the boxes of both children.  As four float4:
//   q0 = lo0.x lo0.y lo0.z hi0.x
//   q1 = hi0.y hi0.z lo1.x lo1.y
//   q2 = lo1.z hi1.x hi1.y hi1.z
//   q3 = ref0 ref1 kind0 kind1           (bit patterns)
struct alignas(16) WideNode {
  float lo0[3];
  float hi0[3];
  float lo1[3];
  float hi1[3];
  uint32_t ref0, ref1;
  uint32_t kind0, kind1;
};
static_assert(sizeof(WideNode) == 64, "WideNode must be 64 bytes");

// Four-child node of the fast traversal's own tree (128 bytes, one cache line), planes grouped so that a
// ray reads its near and far planes directly (which of lo / hi is near depends only on the sign of the
// ray direction on that axis, so a lane addresses piece 2*axis + sign instead of selecting afterwards).
// As eight float4 "pieces": 0 lo.x[4], 1 hi.x[4], 2 lo.y[4], 3 hi.y[4], 4 lo.z[4], 5 hi.z[4], 6 ref[4], 7 unused.
// An unused child slot has an inverted box (lo = +inf, hi = -inf) that no ray enters.
// A child reference with REF_NOCULL set leads to an ill-conditioned triangle (triangle_is_ill_conditioned):
// that child is entered whenever the ray passes its box, however far away the box begins, because such a
// triangle can report a hit at any distance.  (Leaves of this tree hold one or two triangles, so of the three
// count bits of a leaf reference only the lowest is in use.)
static const uint32_t REF_NOCULL = 0x40000000u;
struct alignas(16) Quad4Node {
  float lox[4], hix[4], loy[4], hiy[4], loz[4], hiz[4];
  uint32_t ref[4];
  uint32_t pad[4];
};
static_assert(sizeof(Quad4Node) == 128, "Quad4Node must be 128 bytes");

// Eight-child node of the fast traversal's tree for scenes outside on-chip memory, where a walk's cost is the
// number of node records it fetches one after the other: eight children resolve in one record, so a descent
// takes fewer fetches.  80 bytes in use (five float4 pieces) of a 128-byte slot (one cache line):
//   piece 0  base.x base.y base.z | exps: byte a = biased exponent of the grid step on axis a (step = 2^(byte-127),
//            0 = the node is exactly flat on that axis and every child plane equals base), byte 3 = the axis the
//            children are sorted on (ascending centres)
//   piece 1  child_base | tri_base | masks (bits 0-7: slot holds an internal node, bits 8-15: slot holds a leaf) | -
//   piece 2  x planes as grid indices, one byte per child: lo[0..7] (two words), hi[0..7] (two words)
//   piece 3  y planes, piece 4  z planes
// Children are named by position: the internal children of a node are the nodes child_base, child_base + 1, ... in
// slot order, its leaf children the triangle records tri_base, tri_base + 2, ... (every leaf takes two records; a
// leaf of one triangle is followed by a record no ray hits).  An unused slot is in neither mask.
// A child's plane is base + index * step, rounded outwards by the builder, so the decoded box contains the
// box the four-child tree stores for the same child (which already carries the margin).  A plane the
// four-child tree keeps exactly on a flat gate (see pt_fasttree.cpp) stays exact when the whole node is flat
// there (index 0, plane = base); elsewhere that child gets its unclipped, widened plane instead.
struct alignas(16) Oct8Node {
  float base[3];
  uint32_t exps;
  uint32_t child_base, tri_base, masks, pad0;
  uint8_t qx[16], qy[16], qz[16];   // per axis: lo[8] then hi[8]
  uint32_t pad[12];
};
static_assert(sizeof(Oct8Node) == 128, "Oct8Node must be 128 bytes");

// Box of an internal node of the reference's trees, exactly as given (BVH::Node::box, object.cpp:228-238).
// The fast traversal accepts a triangle hit only if the ray passes the box of the lowest internal node
// above that triangle's leaf ("gate"), tested with AABB::intersect's own arithmetic (object.cpp:179-195).
struct alignas(16) GateBox {
  float lo[3];
  float pad0;
  float hi[3];
  float pad1;
};
static_assert(sizeof(GateBox) == 32, "GateBox must be 32 bytes");
static const uint32_t NO_GATE = 0xffffffffu;

// 48-byte triangle: v0 and the two edges the reference recomputes per test
// (e1 = v1 - v0, e2 = v2 - v0, object.cpp:408-409; the subtraction gives the same
// bits wherever it runs).  Slots are ordered so that a larger slot wins a tie in t.
struct alignas(16) TriRec {
  float v0[3];
  float e1[3];
  float e2[3];
  uint32_t tri_id;   // global triangle id in file order
  uint32_t mat_id;
  uint32_t rank;     // position in the reference's tie order: of two hits at equal t the larger rank wins
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

// Triangle record of the fast tree: the same 48 bytes with the gate index where TriRec keeps the
// triangle id (that one is looked up in a side table when a ray is retired).
struct alignas(16) TriRec4 {
  float v0[3];
  float e1[3];
  float e2[3];
  uint32_t gate;     // index into the gate boxes, or NO_GATE
  uint32_t mat_id;
  uint32_t rank;
};
static_assert(sizeof(TriRec4) == 48, "TriRec4 must be 48 bytes");

struct alignas(16) MatRec {  // Material::diffuse / emission / diffuse_map, object.cpp:146-149
  float kd[3];
  uint32_t tex;  // 0 = no diffuse map, else 1 + index into the texture table
  float ke[3];
  float pad1;
};
static_assert(sizeof(MatRec) == 32, "MatRec must be 32 bytes");

// One diffuse map: `width*height` RGB byte triples starting at byte `offset` of the texel table.
struct alignas(16) TexRec {
  uint32_t offset, width, height, pad;
};
static_assert(sizeof(TexRec) == 16, "TexRec must be 16 bytes");

// One emissive mesh (an entry of Scene::lights(), scene.cpp:22-30).
struct alignas(16) LightRec {
  uint32_t first;     // first record in the light-triangle table
  uint32_t count;
  float area_f;       // (float)Polygon::area(), the upper bound handed to randreal (object.cpp:514)
  float inv_area_f;   // (float)(1 / area), camera.cpp:114
  float ke[3];
  uint32_t mesh_id;
};
static_assert(sizeof(LightRec) == 32, "LightRec must be 32 bytes");

// One triangle of an emissive mesh in creation order (Polygon::triangles), with the
// running sum Polygon::sample walks (object.cpp:515-521).
struct alignas(16) LightTri {
  float u[3];      // v1 - v0
  float v[3];      // v2 - v0
  float n[3];      // Triangle::normal(), un-normalised (object.cpp:393)
  uint32_t pad;
  double accum;    // sum of Triangle::area() up to and including this triangle
};
static_assert(sizeof(LightTri) == 48, "LightTri must be 48 bytes");

}  // namespace ptl
#endif
