// Host-side set-up of the octree (CPU, no device code): amr_build_tree and amr_build_neighbors of the reference
// (octree_mod.f90:460-569, 619-682, 827-849) for hosts that read a list of leaf cells (read_generic_amr.f90) and need
// the cell tables lart_gpu_set_grid_amr takes.  Cell numbers follow the reference: root = 1, a cell is numbered when the
// first leaf that passes through it is inserted, leaves are inserted in input order.
#include "../../include/lart_gpu.h"
#include <cstdint>
#include <string>
#include <vector>

namespace {
thread_local std::string a_err;
int afail(const char *m) { a_err = m; return 1; }
struct Tree {
    std::vector<int32_t> parent, level, ileaf, children, icell_of_leaf;   // children[8*ic + octant]
    std::vector<double> cx, cy, cz, ch;
    int levelmax = 0;
    int add(int par, int lev, double x, double y, double z, double h) {
        parent.push_back(par); level.push_back(lev); ileaf.push_back(0); children.insert(children.end(), 8, 0);
        cx.push_back(x); cy.push_back(y); cz.push_back(z); ch.push_back(h);
        return (int)parent.size();                             // 1-based number of the new cell
    }
};
// leaves in input order, creating the internal cells on the way down (sub-octant bit0 = x >= cx, bit1 = y, bit2 = z)
int build(Tree &t, int32_t nleaf, const double *x, const double *y, const double *z, const int32_t *lev, double lo, double hi) {
    t.add(0, 0, 0.5 * (lo + hi), 0.5 * (lo + hi), 0.5 * (lo + hi), 0.5 * (hi - lo));
    t.icell_of_leaf.assign((size_t)nleaf, 0);
    for (int32_t il = 1; il <= nleaf; il++) {
        if (lev[il - 1] < 0 || lev[il - 1] > 30) return afail("amr_build: leaf level outside 0..30");
        if (x[il - 1] < lo || x[il - 1] > hi || y[il - 1] < lo || y[il - 1] > hi || z[il - 1] < lo || z[il - 1] > hi) return afail("amr_build: leaf centre outside the box");
        int ic = 1;
        for (int l = 0; l < lev[il - 1]; l++) {
            int bx = x[il - 1] >= t.cx[ic - 1], by = y[il - 1] >= t.cy[ic - 1], bz = z[il - 1] >= t.cz[ic - 1], io = bx + 2 * by + 4 * bz;
            if (t.ileaf[ic - 1] > 0) return afail("amr_build: a leaf lies inside another leaf");
            int ch = t.children[8 * (size_t)(ic - 1) + io];
            if (ch == 0) {
                double h = 0.5 * t.ch[ic - 1];
                ch = t.add(ic, l + 1, t.cx[ic - 1] + (2 * bx - 1) * h, t.cy[ic - 1] + (2 * by - 1) * h, t.cz[ic - 1] + (2 * bz - 1) * h, h);
                t.children[8 * (size_t)(ic - 1) + io] = ch;
            }
            ic = ch;
        }
        if (t.ileaf[ic - 1] > 0) return afail("amr_build: two leaves share a cell");
        for (int o = 0; o < 8; o++) if (t.children[8 * (size_t)(ic - 1) + o]) return afail("amr_build: a leaf lies on an internal cell");
        t.ileaf[ic - 1] = il; t.icell_of_leaf[il - 1] = ic;
        if (lev[il - 1] > t.levelmax) t.levelmax = lev[il - 1];
    }
    return 0;
}
// amr_find_cell_at_level: octree_mod.f90:827-849
int find_cell_at_level(const Tree &t, double lo, double hi, double x, double y, double z, int target) {
    if (x < lo || x > hi || y < lo || y > hi || z < lo || z > hi) return 0;
    int ic = 1;
    for (;;) {
        if (t.level[ic - 1] >= target || t.ileaf[ic - 1] > 0) return ic;
        int io = (x >= t.cx[ic - 1]) + 2 * (y >= t.cy[ic - 1]) + 4 * (z >= t.cz[ic - 1]);
        int ch = t.children[8 * (size_t)(ic - 1) + io];
        if (ch == 0) return ic;
        ic = ch;
    }
}
}  // namespace

extern "C" int lart_host_amr_count(int32_t nleaf, const double *x, const double *y, const double *z, const int32_t *level,
                                   double lo, double hi, int32_t *ncells) {
    if (nleaf < 1 || !x || !y || !z || !level || !(hi > lo) || !ncells) return afail("amr_count: bad argument");
    Tree t;
    if (build(t, nleaf, x, y, z, level, lo, hi)) return 1;
    *ncells = (int32_t)t.parent.size();
    return 0;
}

extern "C" int lart_host_amr_build(int32_t nleaf, const double *x, const double *y, const double *z, const int32_t *level,
                                   double lo, double hi, int32_t ncells, double *cx, double *cy, double *cz, double *ch,
                                   int32_t *children, int32_t *neighbor, int32_t *ileaf, int32_t *icell_of_leaf,
                                   int32_t *cell_level, int32_t *levelmax) {
    if (nleaf < 1 || !x || !y || !z || !level || !(hi > lo) || !cx || !cy || !cz || !ch || !children || !neighbor || !ileaf || !icell_of_leaf)
        return afail("amr_build: bad argument");
    Tree t;
    if (build(t, nleaf, x, y, z, level, lo, hi)) return 1;
    if ((int32_t)t.parent.size() != ncells) return afail("amr_build: ncells does not match lart_host_amr_count");
    for (int32_t i = 0; i < ncells; i++) {
        cx[i] = t.cx[i]; cy[i] = t.cy[i]; cz[i] = t.cz[i]; ch[i] = t.ch[i]; ileaf[i] = t.ileaf[i];
        if (cell_level) cell_level[i] = t.level[i];
        for (int o = 0; o < 8; o++) children[8 * (size_t)i + o] = t.children[8 * (size_t)i + o];
    }
    for (int32_t i = 0; i < nleaf; i++) icell_of_leaf[i] = t.icell_of_leaf[i];
    if (levelmax) *levelmax = t.levelmax;
    // amr_build_neighbors: octree_mod.f90:619-682 (faces +x,-x,+y,-y,+z,-z; same-level cell or the coarser cell that covers
    // the position; the outside and an ancestor of the cell itself -- a gap of a sparse tree -- are 0)
    for (int32_t ic = 1; ic <= ncells; ic++) {
        const double px = t.cx[ic - 1], py = t.cy[ic - 1], pz = t.cz[ic - 1], hp = 2.0 * t.ch[ic - 1]; const int lv = t.level[ic - 1];
        int32_t *nb = neighbor + 6 * (size_t)(ic - 1);
        nb[0] = px + hp <= hi ? find_cell_at_level(t, lo, hi, px + hp, py, pz, lv) : 0;
        nb[1] = px - hp >= lo ? find_cell_at_level(t, lo, hi, px - hp, py, pz, lv) : 0;
        nb[2] = py + hp <= hi ? find_cell_at_level(t, lo, hi, px, py + hp, pz, lv) : 0;
        nb[3] = py - hp >= lo ? find_cell_at_level(t, lo, hi, px, py - hp, pz, lv) : 0;
        nb[4] = pz + hp <= hi ? find_cell_at_level(t, lo, hi, px, py, pz + hp, lv) : 0;
        nb[5] = pz - hp >= lo ? find_cell_at_level(t, lo, hi, px, py, pz - hp, lv) : 0;
        for (int f = 0; f < 6; f++) {
            int ng = nb[f];
            if (ng > 0 && ng != ic) { for (int c = t.parent[ic - 1]; c > 0; c = t.parent[c - 1]) if (c == ng) { nb[f] = 0; break; } }
        }
    }
    return 0;
}

extern "C" const char *lart_host_amr_last_error(void) { return a_err.c_str(); }
