// Native LR_ABSCO / LR mesh tokeniser (encode side; host C++, CPU pointer-chasing, not on the decode hot
// path - scope row f4).  Index-based restatement of the reference's half-edge build + traversal:
//   Mesh::Mesh            meto/include/meto/mesh.h:153-262   (discretise, twin edges, boundary marks,
//                                                            half-edge / face ordering, components)
//   Engine_LR_ABSCO::encode / compress_submesh / compress_face
//                         meto/include/meto/engine_lr_absco.h:66-220
// Token alphabet as in meto_decode.h (0 = L, 1 = R, 2 = BOM, 3 + c = coordinate bin c).
//
// The reference orders half-edges and faces with std::sort over comparators that are not strict weak
// orders (two boundary half-edges compare "less" both ways); the result is then whatever the sort
// algorithm does.  To reproduce its streams bit for bit the same std::sort calls are issued here over
// the same sequences with equivalent predicates (same libstdc++ as the reference build in oracle/_ref).
// The recursion of compress_face/compress_submesh is unrolled onto an explicit stack (same visiting order).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <queue>
#include <utility>
#include <vector>

namespace er {

struct MetoMesh {
    struct V { int x, y, z, m; };
    struct HE { int v, s, e, t, n, p, o; };           // opposite / start / end vertex, face, next, prev, twin (-1)
    struct F { int he[3]; int i, ic, m; float cx, cy, cz; };
    std::vector<V> verts;
    std::vector<HE> hes;
    std::vector<F> faces;
    std::vector<int> order;                           // face ids in traversal order

    float twin_dist(int h) const {                    // |v(h) -> v(twin(h))| in grid units, float arithmetic
        const V& a = verts[hes[h].v];
        const V& b = verts[hes[hes[h].o].v];
        const float dx = float(b.x - a.x), dy = float(b.y - a.y), dz = float(b.z - a.z);
        return std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    bool he_less(int a, int b) const {                // HalfEdge::operator<  (mesh.h:113-118)
        if (hes[a].o < 0) return true;
        if (hes[b].o < 0) return false;
        return twin_dist(a) < twin_dist(b);
    }
    bool face_less(int a, int b) const {              // Facet::operator<  (mesh.h:139-143): component, then centre in y-z-x order
        const F &f = faces[a], &g = faces[b];
        if (f.ic != g.ic) return f.ic < g.ic;
        return f.cy < g.cy || (f.cy == g.cy && f.cz < g.cz) || (f.cy == g.cy && f.cz == g.cz && f.cx < g.cx);
    }

    MetoMesh(const float* vtx, int nv, const int32_t* tri, int nf, int bins) {
        verts.resize(nv);
        for (int i = 0; i < nv; ++i) {                // Vertex(float, float, float, bins)  (mesh.h:28-33)
            auto q = [&](float c) { return std::min(int((c + 1) * bins / 2), bins - 1); };
            verts[i] = {q(vtx[3 * i]), q(vtx[3 * i + 1]), q(vtx[3 * i + 2]), 0};
        }
        hes.resize((size_t)3 * nf);
        faces.resize(nf);
        std::map<std::pair<int, int>, int> edge2he;   // -1 = edge already has its two half-edges
        for (int i = 0; i < nf; ++i) {
            F& f = faces[i];
            f.i = i; f.ic = -1; f.m = 0;
            for (int j = 0; j < 3; ++j) {
                const int h = 3 * i + j;
                const int a = tri[3 * i + (j + 1) % 3], b = tri[3 * i + (j + 2) % 3];
                hes[h] = {tri[3 * i + j], a, b, i, -1, -1, -1};
                f.he[j] = h;
                const std::pair<int, int> key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
                auto it = edge2he.find(key);
                if (it == edge2he.end()) {
                    edge2he[key] = h;
                } else if (it->second >= 0) {         // second use of the edge: link twins
                    hes[h].o = it->second;
                    hes[it->second].o = h;
                    it->second = -1;
                }                                     // third+ use: non-manifold, stays a border edge
            }
            for (int j = 0; j < 3; ++j) {
                hes[3 * i + j].n = 3 * i + (j + 1) % 3;
                hes[3 * i + j].p = 3 * i + (j + 2) % 3;
            }
            const V &v0 = verts[tri[3 * i]], &v1 = verts[tri[3 * i + 1]], &v2 = verts[tri[3 * i + 2]];
            f.cx = float(float(v0.x + v1.x + v2.x) / 3.0);
            f.cy = float(float(v0.y + v1.y + v2.y) / 3.0);
            f.cz = float(float(v0.z + v1.z + v2.z) / 3.0);
        }
        for (int i = 0; i < nf; ++i) {
            F& f = faces[i];
            for (int j = 0; j < 3; ++j)
                if (hes[f.he[j]].o < 0) { verts[hes[f.he[j]].s].m = 1; verts[hes[f.he[j]].e].m = 1; }
            std::sort(f.he, f.he + 3, [&](int a, int b) { return he_less(a, b); });
        }
        order.resize(nf);
        for (int i = 0; i < nf; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return face_less(a, b); });
        int ncomp = 0;
        for (int k = 0; k < nf; ++k) {                // connected components in that order (mesh.h:233-258)
            if (faces[order[k]].ic != -1) continue;
            ++ncomp;
            std::queue<int> q;
            q.push(order[k]);
            while (!q.empty()) {
                const int fi = q.front();
                q.pop();
                if (faces[fi].ic != -1) continue;
                faces[fi].ic = ncomp;
                for (int j = 0; j < 3; ++j) {
                    const int o = hes[faces[fi].he[j]].o;
                    if (o >= 0 && faces[hes[o].t].ic == -1) q.push(hes[o].t);
                }
            }
        }
        std::sort(order.begin(), order.end(), [&](int a, int b) { return face_less(a, b); });
    }
};

struct MetoEncodeOut { std::vector<int32_t> tokens, face_order, face_type; };

// RELATIVE = false: LR_ABSCO (absolute coordinates, shorter-boundary-loop heuristic at a split, visited check on deferred
// sub-meshes).  RELATIVE = true: Engine_LR (meto/include/meto/engine_lr.h:59-168): coordinates relative to the previous
// vertex / to the parallelogram prediction v(c) + v(twin) - v(next) - v(prev), offset by bins + 3 (out-of-range -> -1,
// :47-51); a split always walks right first and re-opens the left side as a new sub-mesh WITHOUT checking whether the
// strip just walked has already covered it (:121-127, :130 - so a face can be emitted twice, as in the reference).
template <bool RELATIVE>
inline MetoEncodeOut meto_encode(const float* vtx, int nv, const int32_t* tri, int nf, int bins) {
    enum { OP_L = 0, OP_R = 1, OP_BOM = 2, OP_NUM = 3 };
    MetoMesh M(vtx, nv, tri, nf, bins);
    auto& H = M.hes;
    auto& V = M.verts;
    auto& F = M.faces;
    MetoEncodeOut out;
    auto rel = [&](int d) { return (d < -bins || d >= bins) ? -1 : d + bins + OP_NUM; };
    auto coord = [&](int v) {
        out.tokens.push_back(V[v].x + OP_NUM);
        out.tokens.push_back(V[v].y + OP_NUM);
        out.tokens.push_back(V[v].z + OP_NUM);
    };
    auto coord_delta = [&](int v, int from) {            // v - from (from < 0: absolute)
        const int fx = from < 0 ? 0 : V[from].x, fy = from < 0 ? 0 : V[from].y, fz = from < 0 ? 0 : V[from].z;
        out.tokens.push_back(rel(V[v].x - fx));
        out.tokens.push_back(rel(V[v].y - fy));
        out.tokens.push_back(rel(V[v].z - fz));
    };
    auto face_visited = [&](int h) { return h < 0 || F[H[h].t].m != 0; };   // "o == NULL || o->t->m"
    std::vector<int> pending;                         // sub-meshes whose traversal was deferred at a split (LIFO = recursion order)
    for (int k = 0; k < nf; ++k) {
        if (F[M.order[k]].m) continue;
        pending.push_back(F[M.order[k]].he[0]);
        while (!pending.empty()) {
            int c = pending.back();
            pending.pop_back();
            if (!RELATIVE && F[H[c].t].m) continue;   // compress_submesh: already visited (hole / handle); LR has no such check
            out.tokens.push_back(OP_BOM);
            if (RELATIVE) { coord_delta(H[c].v, -1); coord_delta(H[c].s, H[c].v); coord_delta(H[c].e, H[c].s); }
            else { coord(H[c].v); coord(H[c].s); coord(H[c].e); }
            V[H[c].s].m = 1; V[H[c].e].m = 1;
            bool init = true;
            for (;;) {                                // compress_face chain
                F[H[c].t].m = 1;
                out.face_order.push_back(F[H[c].t].i);
                if (!init) {
                    const int o = H[c].o;
                    if (!(H[c].s == H[o].e && H[c].e == H[o].s)) {          // inconsistent winding: flip this face
                        for (int j = 0; j < 3; ++j) {
                            auto& e = H[F[H[c].t].he[j]];
                            std::swap(e.s, e.e);
                            std::swap(e.n, e.p);
                        }
                    }
                    if (RELATIVE) {               // parallelogram correction (twin read after a possible flip, like the reference)
                        const int vo = H[H[c].o].v, vn = H[H[c].n].v, vp = H[H[c].p].v;
                        out.tokens.push_back(rel(V[H[c].v].x + V[vo].x - V[vn].x - V[vp].x));
                        out.tokens.push_back(rel(V[H[c].v].y + V[vo].y - V[vn].y - V[vp].y));
                        out.tokens.push_back(rel(V[H[c].v].z + V[vo].z - V[vn].z - V[vp].z));
                    } else {
                        coord(H[c].v);
                    }
                }
                init = false;
                const bool tip = V[H[c].v].m != 0;
                const int left = H[H[c].p].o, right = H[H[c].n].o;
                const bool left_v = face_visited(left), right_v = face_visited(right);
                if (!tip) {                           // new vertex: "C" merged into L
                    V[H[c].v].m = 1;
                    out.tokens.push_back(OP_L); out.face_type.push_back(OP_L);
                    c = right;
                } else if (left_v && right_v) {       // "E": end of this strip
                    out.face_type.push_back(OP_BOM);
                    break;
                } else if (left_v) {
                    out.tokens.push_back(OP_L); out.face_type.push_back(OP_L);
                    c = right;
                } else if (right_v) {
                    out.tokens.push_back(OP_R); out.face_type.push_back(OP_R);
                    c = left;
                } else if (RELATIVE) {                // "S" in Engine_LR: right strip first, left side re-opened afterwards
                    out.tokens.push_back(OP_L); out.face_type.push_back(OP_L);
                    pending.push_back(left);
                    c = right;
                } else {                              // "S": split - walk both unvisited boundary loops, shorter side first
                    int len_left = 0, len_right = 0;
                    for (int cur = right;;) {
                        ++len_left;
                        cur = H[cur].n;
                        while (H[cur].o >= 0 && !F[H[H[cur].o].t].m) cur = H[H[cur].o].n;
                        if (cur == right) break;
                    }
                    for (int cur = left;;) {
                        ++len_right;
                        cur = H[cur].p;
                        while (H[cur].o >= 0 && !F[H[H[cur].o].t].m) cur = H[H[cur].o].p;
                        if (cur == left) break;
                    }
                    if (len_left < len_right) {
                        out.tokens.push_back(OP_L); out.face_type.push_back(OP_L);
                        pending.push_back(left);
                        c = right;
                    } else {
                        out.tokens.push_back(OP_R); out.face_type.push_back(OP_R);
                        pending.push_back(right);
                        c = left;
                    }
                }
            }
        }
    }
    return out;
}

}  // namespace er
