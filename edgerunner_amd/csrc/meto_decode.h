// Native detokeniser for the LR_ABSCO mesh token stream (host C++; a linear scan, CPU-bound,
// runs once per generated sample after the decode loop).  Behaviour follows the reference's
// only native component: Engine_LR_ABSCO::decode (meto/include/meto/engine_lr_absco.h:223-295)
// with Vertex::undiscrete (meto/include/meto/mesh.h:36-42).  Token alphabet (meto ids =
// model ids - 3): 0 = L, 1 = R, 2 = BOM, 3 + c = coordinate bin c.
//
// Stream grammar: BOM x0 y0 z0 x1 y1 z1 x2 y2 z2 opens a sub-mesh with one triangle
// (v0,v1,v2); then each (L|R) x y z adds one vertex v and one triangle sharing an edge with
// the previous one:  L -> (v, v0, v2), then v1 <- v0, v0 <- v
//                    R -> (v, v1, v0), then v2 <- v0, v0 <- v.
// A truncated group or a coordinate where an op is expected stops the scan (as the reference does).
#pragma once
#include <stdint.h>

namespace er {

struct MetoCounts { int vertices, faces, face_types; };

inline MetoCounts meto_decode_lr_absco(const int32_t* tok, int n, int bins, float* vout, int32_t* fout, int32_t* tout) {
    enum { OP_L = 0, OP_R = 1, OP_BOM = 2, OP_NUM = 3 };
    int nv = 0, nf = 0, nt = 0;
    int i0 = -1, i1 = -1, i2 = -1;            // indices of the rolling window v0, v1, v2
    auto emit_vertex = [&](int a, int b, int c) {
        const int q[3] = {tok[a] - OP_NUM, tok[b] - OP_NUM, tok[c] - OP_NUM};
        for (int k = 0; k < 3; ++k)            // float((float(x) + 0.5) / bins * 2 - 1): double arithmetic, one final rounding
            vout[3 * nv + k] = (float)(((double)(float)q[k] + 0.5) / bins * 2 - 1);
        return nv++;
    };
    for (int i = 0; i < n; ++i) {
        if (tok[i] == OP_BOM) {
            if (i + 9 >= n) break;             // incomplete group: all nine coordinates i+1..i+9 must exist
            i0 = emit_vertex(i + 1, i + 2, i + 3);
            i1 = emit_vertex(i + 4, i + 5, i + 6);
            i2 = emit_vertex(i + 7, i + 8, i + 9);
            fout[3 * nf] = i0; fout[3 * nf + 1] = i1; fout[3 * nf + 2] = i2; ++nf;
            if (i != 0) tout[nt++] = OP_BOM;
            i += 9;
        } else {
            if (tok[i] >= OP_NUM) break;       // a coordinate where an op must be
            if (i + 3 >= n) break;
            const int op = tok[i];
            if (op == OP_L) {
                const int v = emit_vertex(i + 1, i + 2, i + 3);
                fout[3 * nf] = v; fout[3 * nf + 1] = i0; fout[3 * nf + 2] = i2; ++nf;
                i1 = i0; i0 = v;
            } else if (op == OP_R) {
                const int v = emit_vertex(i + 1, i + 2, i + 3);
                fout[3 * nf] = v; fout[3 * nf + 1] = i1; fout[3 * nf + 2] = i0; ++nf;
                i2 = i0; i0 = v;
            }                                   // negative ids (PAD/BOS/EOS shifted by -3) fall through like the reference
            tout[nt++] = op;
            i += 3;
        }
    }
    tout[nt++] = OP_BOM;                        // "last face" marker
    return {nv, nf, nt};
}

// The LR backend (Options.meto_backend = 'LR'): same stream shape, RELATIVE coordinates with parallelogram
// prediction; Engine_LR::decode (meto/include/meto/engine_lr.h:171-253).  A coordinate token t stands for
// t - bins - 3 (restore_coord, :53-56; negative tokens pass through unchanged).  BOM: v0 absolute, v1 = v0 + d,
// v2 = v1 + d; L: v = v0 + v2 - v1 + d; R: v = v0 + v1 - v2 + d; window updates as in LR_ABSCO.
inline MetoCounts meto_decode_lr(const int32_t* tok, int n, int bins, float* vout, int32_t* fout, int32_t* tout) {
    enum { OP_L = 0, OP_R = 1, OP_BOM = 2, OP_NUM = 3 };
    struct P { int x, y, z, i; };
    int nv = 0, nf = 0, nt = 0;
    P v0{0, 0, 0, -1}, v1{0, 0, 0, -1}, v2{0, 0, 0, -1};
    auto restore = [&](int t) { return t < 0 ? t : t - bins - OP_NUM; };
    auto emit = [&](P& p) {
        const int q[3] = {p.x, p.y, p.z};
        for (int k = 0; k < 3; ++k) vout[3 * nv + k] = (float)(((double)(float)q[k] + 0.5) / bins * 2 - 1);
        p.i = nv++;
    };
    for (int i = 0; i < n; ++i) {
        if (tok[i] == OP_BOM) {
            if (i + 9 >= n) break;
            v0 = {restore(tok[i + 1]), restore(tok[i + 2]), restore(tok[i + 3]), -1};
            v1 = {v0.x + restore(tok[i + 4]), v0.y + restore(tok[i + 5]), v0.z + restore(tok[i + 6]), -1};
            v2 = {v1.x + restore(tok[i + 7]), v1.y + restore(tok[i + 8]), v1.z + restore(tok[i + 9]), -1};
            emit(v0); emit(v1); emit(v2);
            fout[3 * nf] = v0.i; fout[3 * nf + 1] = v1.i; fout[3 * nf + 2] = v2.i; ++nf;
            if (i != 0) tout[nt++] = OP_BOM;
            i += 9;
        } else {
            if (tok[i] >= OP_NUM) break;
            if (i + 3 >= n) break;
            const int op = tok[i];
            const int dx = restore(tok[i + 1]), dy = restore(tok[i + 2]), dz = restore(tok[i + 3]);
            if (op == OP_L) {
                P v{v0.x + v2.x - v1.x + dx, v0.y + v2.y - v1.y + dy, v0.z + v2.z - v1.z + dz, -1};
                emit(v);
                fout[3 * nf] = v.i; fout[3 * nf + 1] = v0.i; fout[3 * nf + 2] = v2.i; ++nf;
                v1 = v0; v0 = v;
            } else if (op == OP_R) {
                P v{v0.x + v1.x - v2.x + dx, v0.y + v1.y - v2.y + dy, v0.z + v1.z - v2.z + dz, -1};
                emit(v);
                fout[3 * nf] = v.i; fout[3 * nf + 1] = v1.i; fout[3 * nf + 2] = v0.i; ++nf;
                v2 = v0; v0 = v;
            }
            tout[nt++] = op;
            i += 3;
        }
    }
    tout[nt++] = OP_BOM;
    return {nv, nf, nt};
}

}  // namespace er
