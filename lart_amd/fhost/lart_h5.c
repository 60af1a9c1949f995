/* Output-container helpers of the Fortran host: the HDF5 layout of the reference (hdf5io_mod.f90:8-13: one group per
 * section named by its EXTNAME, image sections hold a dataset `data`, table sections one 1-D dataset per column,
 * keywords are attributes of the group; Fortran array order = reversed HDF5 dimensions) through the HDF5 C API, and gzip
 * for the FITS form (<base>.fits.gz).  Built with -DLART_HDF5 when <hdf5.h> is in the image; otherwise the HDF5 entry
 * points report "unavailable" and the host writes FITS. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <zlib.h>

#ifdef LART_HDF5
#include <hdf5.h>
int lh5_available(void) { return 1; }
long long lh5_create(const char *path) {
    hid_t fcpl = H5Pcreate(H5P_FILE_CREATE);
    H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);      /* sections keep their writing order */
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, fcpl, H5P_DEFAULT);
    H5Pclose(fcpl);
    return (long long)f;
}
long long lh5_group(long long file, const char *name) {
    hid_t gcpl = H5Pcreate(H5P_GROUP_CREATE);
    H5Pset_link_creation_order(gcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    hid_t g = H5Gcreate2((hid_t)file, name, H5P_DEFAULT, gcpl, H5P_DEFAULT);
    H5Pclose(gcpl);
    return (long long)g;
}
/* par%out_bitpix (define.f90:418): -32 stores the datasets as float32, -64 (and 0 here: this host keeps full precision where the
   reference's auto mode may pick float32) as float64; the memory type stays double, the library converts */
static int g_bitpix = -64;
void lh5_set_bitpix(int bp) { g_bitpix = (bp == -32) ? -32 : -64; }
/* fdims = Fortran extents (fastest first); the file gets them reversed, as the HDF5 Fortran API does */
int lh5_put_f64(long long grp, const char *name, int rank, const long long *fdims, const double *a) {
    hsize_t d[8]; if (rank < 1 || rank > 8) return 1;
    for (int i = 0; i < rank; i++) d[i] = (hsize_t)fdims[rank - 1 - i];
    hid_t sp = H5Screate_simple(rank, d, NULL), ds = H5Dcreate2((hid_t)grp, name, g_bitpix == -32 ? H5T_IEEE_F32LE : H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    int rc = (ds < 0) || H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a) < 0;
    if (ds >= 0) H5Dclose(ds);
    H5Sclose(sp);
    return rc;
}
static int put_attr(long long grp, const char *name, hid_t type, const void *v) {
    hid_t sp = H5Screate(H5S_SCALAR), at = H5Acreate2((hid_t)grp, name, type, sp, H5P_DEFAULT, H5P_DEFAULT);
    int rc = (at < 0) || H5Awrite(at, type, v) < 0;
    if (at >= 0) H5Aclose(at);
    H5Sclose(sp);
    return rc;
}
int lh5_attr_f64(long long grp, const char *name, double v) { return put_attr(grp, name, H5T_NATIVE_DOUBLE, &v); }
int lh5_attr_i32(long long grp, const char *name, int v) { return put_attr(grp, name, H5T_NATIVE_INT, &v); }
int lh5_attr_str(long long grp, const char *name, const char *v) {
    hid_t t = H5Tcopy(H5T_C_S1); size_t n = strlen(v); H5Tset_size(t, n > 0 ? n : 1);
    int rc = put_attr(grp, name, t, n > 0 ? v : " "); H5Tclose(t);
    return rc;
}
int lh5_close_group(long long grp) { return H5Gclose((hid_t)grp) < 0; }
int lh5_close(long long file) { return H5Fclose((hid_t)file) < 0; }
#else
int lh5_available(void) { return 0; }
void lh5_set_bitpix(int bp) { (void)bp; }
long long lh5_create(const char *path) { (void)path; return -1; }
long long lh5_group(long long file, const char *name) { (void)file; (void)name; return -1; }
int lh5_put_f64(long long grp, const char *name, int rank, const long long *fdims, const double *a) { (void)grp; (void)name; (void)rank; (void)fdims; (void)a; return 1; }
int lh5_attr_f64(long long grp, const char *name, double v) { (void)grp; (void)name; (void)v; return 1; }
int lh5_attr_i32(long long grp, const char *name, int v) { (void)grp; (void)name; (void)v; return 1; }
int lh5_attr_str(long long grp, const char *name, const char *v) { (void)grp; (void)name; (void)v; return 1; }
int lh5_close_group(long long grp) { (void)grp; return 1; }
int lh5_close(long long file) { (void)file; return 1; }
#endif

/* <path> -> <path>.gz (the reference writes gzip FITS through CFITSIO); removes <path> on success */
int lart_gzip_file(const char *path) {
    char out[1024]; if (snprintf(out, sizeof out, "%s.gz", path) >= (int)sizeof out) return 1;
    FILE *f = fopen(path, "rb"); if (!f) return 1;
    gzFile g = gzopen(out, "wb6"); if (!g) { fclose(f); return 1; }
    static char buf[1 << 20]; size_t n; int rc = 0;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) if (gzwrite(g, buf, (unsigned)n) != (int)n) { rc = 1; break; }
    fclose(f); if (gzclose(g) != Z_OK) rc = 1;
    if (!rc) remove(path);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Generic AMR leaf tables in the binary containers of the reference (read_generic_amr.f90:186-279; written by
 * python/AMR_grid/AMR_grid.py:1544-1630): a FITS binary table (.fits / .fits.gz, second HDU) or an HDF5 group of 1-D
 * datasets.  Columns 1-9 = x y z level dens T vx vy vz; keywords BOXLEN (mandatory) and ORIGINX/Y/Z (default -BOXLEN/2).
 * data = 9 * nleaf doubles, column after column (malloc'd here, released with lart_free).
 * ------------------------------------------------------------------------------------------------------------------ */
void lart_free(void *p) { free(p); }

static double be_f64(const unsigned char *b) { unsigned long long v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | b[i]; double d; memcpy(&d, &v, 8); return d; }
static double be_f32(const unsigned char *b) { unsigned v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | b[i]; float f; memcpy(&f, &v, 4); return (double)f; }
static double be_int(const unsigned char *b, int n) { long long v = (b[0] & 0x80) ? -1 : 0; for (int i = 0; i < n; i++) v = (v << 8) | b[i]; return (double)v; }

/* one FITS header: calls back with every card until END; returns 0, or 1 at end of file */
static int fits_header(gzFile f, void (*card)(const char *key, const char *val, void *u), void *u) {
    char blk[2880]; int done = 0;
    while (!done) {
        if (gzread(f, blk, 2880) != 2880) return 1;
        for (int c = 0; c < 36 && !done; c++) {
            char key[9], val[72]; memcpy(key, blk + 80 * c, 8); key[8] = 0;
            for (int i = 7; i >= 0 && key[i] == ' '; i--) key[i] = 0;
            if (!strcmp(key, "END")) { done = 1; break; }
            if (blk[80 * c + 8] != '=') continue;
            const char *p = blk + 80 * c + 10; int n = 0;
            while (p < blk + 80 * c + 80 && *p == ' ') p++;
            if (*p == '\'') { p++; while (p < blk + 80 * c + 80 && *p != '\'' && n < 70) val[n++] = *p++; while (n > 0 && val[n-1] == ' ') n--; }
            else { while (p < blk + 80 * c + 80 && *p != '/' && *p != ' ' && n < 70) val[n++] = *p++; }
            val[n] = 0;
            card(key, val, u);
        }
    }
    return 0;
}
typedef struct { long naxis, naxes[8], bitpix, tfields, pcount; char tform[64][12]; double boxlen, origin[3]; int has_origin[3], bintable; } fits_hdr;
static void fits_card(const char *k, const char *v, void *u) {
    fits_hdr *h = (fits_hdr *)u;
    if (!strcmp(k, "NAXIS")) h->naxis = atol(v);
    else if (!strncmp(k, "NAXIS", 5) && k[5] >= '1' && k[5] <= '8' && !k[6]) h->naxes[k[5] - '1'] = atol(v);
    else if (!strcmp(k, "BITPIX")) h->bitpix = atol(v);
    else if (!strcmp(k, "PCOUNT")) h->pcount = atol(v);
    else if (!strcmp(k, "TFIELDS")) h->tfields = atol(v);
    else if (!strcmp(k, "XTENSION")) h->bintable = !strncmp(v, "BINTABLE", 8);
    else if (!strncmp(k, "TFORM", 5)) { int i = atoi(k + 5); if (i >= 1 && i <= 64) { strncpy(h->tform[i-1], v, 11); h->tform[i-1][11] = 0; } }
    else if (!strcmp(k, "BOXLEN")) h->boxlen = atof(v);
    else if (!strcmp(k, "ORIGINX")) { h->origin[0] = atof(v); h->has_origin[0] = 1; }
    else if (!strcmp(k, "ORIGINY")) { h->origin[1] = atof(v); h->has_origin[1] = 1; }
    else if (!strcmp(k, "ORIGINZ")) { h->origin[2] = atof(v); h->has_origin[2] = 1; }
}
static int amr_read_fits(const char *path, int *nleaf, double *boxlen, double *origin, double **data) {
    gzFile f = gzopen(path, "rb"); if (!f) return 1;                    /* (zlib reads plain files through the same calls) */
    fits_hdr h; memset(&h, 0, sizeof h);
    if (fits_header(f, fits_card, &h)) { gzclose(f); return 2; }
    long long skip = 0;
    if (h.naxis > 0) { skip = labs(h.bitpix) / 8; for (long i = 0; i < h.naxis; i++) skip *= h.naxes[i]; skip = (skip + 2879) / 2880 * 2880; }
    char junk[2880]; for (long long s = 0; s < skip; s += 2880) if (gzread(f, junk, 2880) != 2880) { gzclose(f); return 2; }
    memset(&h, 0, sizeof h);
    if (fits_header(f, fits_card, &h) || !h.bintable || h.tfields < 9 || h.naxes[1] <= 0 || !(h.boxlen > 0.0)) { gzclose(f); return 3; }
    int width[64], off[64], kind[64], row = 0;                          /* kind: 0 f64, 1 f32, 2 int */
    for (long i = 0; i < h.tfields && i < 64; i++) {
        const char *t = h.tform[i]; int rep = 1; while (*t >= '0' && *t <= '9') { rep = (t == h.tform[i]) ? (*t - '0') : rep * 10 + (*t - '0'); t++; }
        int w = 0, k = 0;
        switch (*t) { case 'D': w = 8; k = 0; break; case 'E': w = 4; k = 1; break; case 'J': w = 4; k = 2; break; case 'K': w = 8; k = 2; break;
                      case 'I': w = 2; k = 2; break; case 'B': case 'L': case 'A': w = 1; k = 2; break; default: gzclose(f); return 4; }
        if (i < 9 && rep != 1) { gzclose(f); return 4; }
        off[i] = row; width[i] = w; kind[i] = k; row += w * rep;
    }
    if (row != h.naxes[0]) { gzclose(f); return 4; }
    long n = h.naxes[1];
    double *d = (double *)malloc(sizeof(double) * 9 * (size_t)n); unsigned char *buf = (unsigned char *)malloc((size_t)row);
    if (!d || !buf) { free(d); free(buf); gzclose(f); return 5; }
    for (long r = 0; r < n; r++) {
        if (gzread(f, buf, (unsigned)row) != row) { free(d); free(buf); gzclose(f); return 2; }
        for (int c = 0; c < 9; c++) { const unsigned char *b = buf + off[c];
            d[(size_t)c * n + r] = kind[c] == 0 ? be_f64(b) : kind[c] == 1 ? be_f32(b) : be_int(b, width[c]); }
    }
    free(buf); gzclose(f);
    *nleaf = (int)n; *boxlen = h.boxlen; for (int a = 0; a < 3; a++) origin[a] = h.has_origin[a] ? h.origin[a] : -0.5 * h.boxlen;
    *data = d;
    return 0;
}

#ifdef LART_HDF5
static int h5_attr_f64(hid_t g, const char *name, double *v) {
    if (H5Aexists(g, name) <= 0) return 1;
    hid_t a = H5Aopen(g, name, H5P_DEFAULT); if (a < 0) return 1;
    int rc = H5Aread(a, H5T_NATIVE_DOUBLE, v) < 0; H5Aclose(a); return rc;
}
static int amr_read_h5(const char *path, int *nleaf, double *boxlen, double *origin, double **data) {
    hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT); if (f < 0) return 1;
    char gname[256]; gname[0] = 0;
    H5G_info_t gi; if (H5Gget_info(f, &gi) < 0 || gi.nlinks < 1) { H5Fclose(f); return 3; }
    for (hsize_t i = 0; i < gi.nlinks && !gname[0]; i++) {              /* the first group of the file */
        char nm[256]; if (H5Lget_name_by_idx(f, ".", H5_INDEX_NAME, H5_ITER_INC, i, nm, sizeof nm, H5P_DEFAULT) < 0) continue;
        hid_t o = H5Oopen(f, nm, H5P_DEFAULT);                         /* (H5Oget_info changed its signature between 1.10 and 1.12) */
        if (o >= 0) { if (H5Iget_type(o) == H5I_GROUP) strcpy(gname, nm); H5Oclose(o); }
    }
    if (!gname[0]) { H5Fclose(f); return 3; }
    hid_t g = H5Gopen2(f, gname, H5P_DEFAULT);
    if (h5_attr_f64(g, "BOXLEN", boxlen) || !(*boxlen > 0.0)) { H5Gclose(g); H5Fclose(f); return 3; }
    const char *on[3] = {"ORIGINX", "ORIGINY", "ORIGINZ"};
    for (int a = 0; a < 3; a++) if (h5_attr_f64(g, on[a], &origin[a])) origin[a] = -0.5 * (*boxlen);
    const char *cols[9][3] = {{"x", 0, 0}, {"y", 0, 0}, {"z", 0, 0}, {"level", 0, 0}, {"dens", "nH", "density"}, {"T", "temperature", 0},
                              {"vx", 0, 0}, {"vy", 0, 0}, {"vz", 0, 0}};
    double *d = NULL; hsize_t n = 0;
    for (int c = 0; c < 9; c++) {
        hid_t ds = -1;
        for (int k = 0; k < 3 && ds < 0; k++) if (cols[c][k] && H5Lexists(g, cols[c][k], H5P_DEFAULT) > 0) ds = H5Dopen2(g, cols[c][k], H5P_DEFAULT);
        if (ds < 0) { free(d); H5Gclose(g); H5Fclose(f); return 4; }
        hid_t sp = H5Dget_space(ds); hsize_t dims[4] = {0, 0, 0, 0}; int rk = H5Sget_simple_extent_dims(sp, dims, NULL); H5Sclose(sp);
        if (rk != 1 || (c > 0 && dims[0] != n)) { H5Dclose(ds); free(d); H5Gclose(g); H5Fclose(f); return 4; }
        if (c == 0) { n = dims[0]; d = (double *)malloc(sizeof(double) * 9 * (size_t)n); if (!d) { H5Dclose(ds); H5Gclose(g); H5Fclose(f); return 5; } }
        int rc = H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, d + (size_t)c * n) < 0;       /* any stored type converts */
        H5Dclose(ds);
        if (rc) { free(d); H5Gclose(g); H5Fclose(f); return 2; }
    }
    H5Gclose(g); H5Fclose(f);
    *nleaf = (int)n; *data = d;
    return 0;
}
#else
static int amr_read_h5(const char *path, int *nleaf, double *boxlen, double *origin, double **data) { (void)path; (void)nleaf; (void)boxlen; (void)origin; (void)data; return 6; }
#endif

/* 0 ok, 1 cannot open, 2 short read, 3 not a leaf table (no BINTABLE / group / BOXLEN), 4 unexpected columns, 5 memory, 6 no HDF5 in this build */
int lart_amr_read_binary(const char *path, int *nleaf, double *boxlen, double *origin, double **data) {
    size_t L = strlen(path);
    if ((L > 3 && !strcmp(path + L - 3, ".h5")) || (L > 5 && !strcmp(path + L - 5, ".hdf5"))) return amr_read_h5(path, nleaf, boxlen, origin, data);
    return amr_read_fits(path, nleaf, boxlen, origin, data);
}


/* ---- clump population files (write_clumps_info / read_clumps_info, clump_mod.f90:1779-1961, 2040-2312): the first table section of a
 * FITS[.gz] file (BINTABLE after the primary HDU) or of an HDF5 file (the first group with a dataset X).  data = 9 * n doubles, column after
 * column: X Y Z VX VY VZ R_CLUMP RHOKAP TEMP; present[c] = 0 when the file has no such column (RHOKAP: 1 = RHOKAP, 2 = DENSITY, 3 = DENS,
 * read_perclump_or_keyword :1965-2036).  keys = N_CLUMPS CL_RAD RHOKAP TEMP_CL DIST_CM (NaN when absent), distunit = DISTUNIT ('' when absent).
 * 0 ok, 1 cannot open, 2 short read, 3 no table section, 4 unexpected layout, 5 memory, 6 no HDF5 in this build */
static const char *cl_col[9][3] = {{"X", 0, 0}, {"Y", 0, 0}, {"Z", 0, 0}, {"VX", 0, 0}, {"VY", 0, 0}, {"VZ", 0, 0}, {"R_CLUMP", 0, 0}, {"RHOKAP", "DENSITY", "DENS"}, {"TEMP", 0, 0}};
static const char *cl_key[5] = {"N_CLUMPS", "CL_RAD", "RHOKAP", "TEMP_CL", "DIST_CM"};
typedef struct { fits_hdr h; char ttype[64][24]; double keys[5]; int has[5]; char distunit[64]; } cl_hdr;
static void cl_card(const char *k, const char *v, void *u) {
    cl_hdr *c = (cl_hdr *)u; fits_card(k, v, &c->h);
    if (!strncmp(k, "TTYPE", 5)) { int i = atoi(k + 5); if (i >= 1 && i <= 64) { strncpy(c->ttype[i-1], v, 23); c->ttype[i-1][23] = 0; } }
    for (int q = 0; q < 5; q++) if (!strcmp(k, cl_key[q])) { c->keys[q] = atof(v); c->has[q] = 1; }
    if (!strcmp(k, "DISTUNIT")) { strncpy(c->distunit, v, 63); c->distunit[63] = 0; }
}
static int clumps_read_fits(const char *path, long long *n_out, double **data, int *present, double *keys, char *distunit) {
    gzFile f = gzopen(path, "rb"); if (!f) return 1;
    cl_hdr c; memset(&c, 0, sizeof c);
    if (fits_header(f, cl_card, &c)) { gzclose(f); return 2; }
    long long skip = 0;
    if (c.h.naxis > 0) { skip = labs(c.h.bitpix) / 8; for (long i = 0; i < c.h.naxis; i++) skip *= c.h.naxes[i]; skip = (skip + 2879) / 2880 * 2880; }
    char junk[2880]; for (long long s = 0; s < skip; s += 2880) if (gzread(f, junk, 2880) != 2880) { gzclose(f); return 2; }
    memset(&c, 0, sizeof c);
    if (fits_header(f, cl_card, &c) || !c.h.bintable || c.h.tfields < 1 || c.h.naxes[1] <= 0) { gzclose(f); return 3; }
    int width[64], off[64], kind[64], row = 0;
    for (long i = 0; i < c.h.tfields && i < 64; i++) {
        const char *t = c.h.tform[i]; int rep = 0; while (*t >= '0' && *t <= '9') { rep = rep * 10 + (*t - '0'); t++; } if (rep == 0) rep = 1;
        int w = 0, k = 0;
        switch (*t) { case 'D': w = 8; k = 0; break; case 'E': w = 4; k = 1; break; case 'J': w = 4; k = 2; break; case 'K': w = 8; k = 2; break;
                      case 'I': w = 2; k = 2; break; case 'B': case 'L': case 'A': w = 1; k = 2; break; default: gzclose(f); return 4; }
        off[i] = row; width[i] = w; kind[i] = k; row += w * rep;
    }
    if (row != c.h.naxes[0]) { gzclose(f); return 4; }
    int src[9];
    for (int q = 0; q < 9; q++) { src[q] = -1; present[q] = 0;
        for (int a = 0; a < 3 && src[q] < 0; a++) if (cl_col[q][a]) for (long i = 0; i < c.h.tfields && i < 64; i++) if (!strcmp(c.ttype[i], cl_col[q][a])) { src[q] = (int)i; present[q] = a + 1; break; } }
    for (int q = 0; q < 6; q++) if (src[q] < 0) { gzclose(f); return 4; }        /* positions and velocities are columns of every population file (the HDF5 reader fails the same way) */
    long n = c.h.naxes[1];
    double *d = (double *)calloc((size_t)9 * (size_t)n, sizeof(double)); unsigned char *buf = (unsigned char *)malloc((size_t)row);
    if (!d || !buf) { free(d); free(buf); gzclose(f); return 5; }
    for (long r = 0; r < n; r++) {
        if (gzread(f, buf, (unsigned)row) != row) { free(d); free(buf); gzclose(f); return 2; }
        for (int q = 0; q < 9; q++) if (src[q] >= 0) { const unsigned char *b = buf + off[src[q]]; int kk = kind[src[q]];
            d[(size_t)q * n + r] = kk == 0 ? be_f64(b) : kk == 1 ? be_f32(b) : be_int(b, width[src[q]]); }
    }
    free(buf); gzclose(f);
    for (int q = 0; q < 5; q++) keys[q] = c.has[q] ? c.keys[q] : NAN;
    strcpy(distunit, c.distunit);
    *n_out = n; *data = d;
    return 0;
}
#ifdef LART_HDF5
static int clumps_read_h5(const char *path, long long *n_out, double **data, int *present, double *keys, char *distunit) {
    hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT); if (f < 0) return 1;
    H5G_info_t gi; if (H5Gget_info(f, &gi) < 0 || gi.nlinks < 1) { H5Fclose(f); return 3; }
    hid_t g = -1;
    for (hsize_t i = 0; i < gi.nlinks && g < 0; i++) {
        char nm[256]; if (H5Lget_name_by_idx(f, ".", H5_INDEX_NAME, H5_ITER_INC, i, nm, sizeof nm, H5P_DEFAULT) < 0) continue;
        hid_t o = H5Oopen(f, nm, H5P_DEFAULT);
        if (o >= 0) { if (H5Iget_type(o) == H5I_GROUP && H5Lexists(o, "X", H5P_DEFAULT) > 0) g = H5Gopen2(f, nm, H5P_DEFAULT); H5Oclose(o); }
    }
    if (g < 0) { H5Fclose(f); return 3; }
    double *d = NULL; hsize_t n = 0;
    for (int q = 0; q < 9; q++) {
        hid_t ds = -1; present[q] = 0;
        for (int a = 0; a < 3 && ds < 0; a++) if (cl_col[q][a] && H5Lexists(g, cl_col[q][a], H5P_DEFAULT) > 0) { ds = H5Dopen2(g, cl_col[q][a], H5P_DEFAULT); present[q] = a + 1; }
        if (ds < 0) { if (q < 6) { free(d); H5Gclose(g); H5Fclose(f); return 4; } continue; }
        hid_t sp = H5Dget_space(ds); hsize_t dims[4] = {0, 0, 0, 0}; int rk = H5Sget_simple_extent_dims(sp, dims, NULL); H5Sclose(sp);
        if (rk != 1 || (q > 0 && dims[0] != n)) { H5Dclose(ds); free(d); H5Gclose(g); H5Fclose(f); return 4; }
        if (q == 0) { n = dims[0]; d = (double *)calloc((size_t)9 * (size_t)n, sizeof(double)); if (!d) { H5Dclose(ds); H5Gclose(g); H5Fclose(f); return 5; } }
        int rc = H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, d + (size_t)q * n) < 0; H5Dclose(ds);
        if (rc) { free(d); H5Gclose(g); H5Fclose(f); return 2; }
    }
    for (int q = 0; q < 5; q++) if (h5_attr_f64(g, cl_key[q], &keys[q])) keys[q] = NAN;
    distunit[0] = 0;
    if (H5Aexists(g, "DISTUNIT") > 0) { hid_t a = H5Aopen(g, "DISTUNIT", H5P_DEFAULT); hid_t t = H5Aget_type(a);
        if (H5Tget_class(t) == H5T_STRING && !H5Tis_variable_str(t) && H5Tget_size(t) < 63) { char b[64]; memset(b, 0, sizeof b); if (H5Aread(a, t, b) >= 0) { strcpy(distunit, b); for (int i = (int)strlen(distunit) - 1; i >= 0 && distunit[i] == ' '; i--) distunit[i] = 0; } }
        H5Tclose(t); H5Aclose(a); }
    H5Gclose(g); H5Fclose(f);
    *n_out = (long long)n; *data = d;
    return 0;
}
#else
static int clumps_read_h5(const char *path, long long *n_out, double **data, int *present, double *keys, char *distunit) { (void)path; (void)n_out; (void)data; (void)present; (void)keys; (void)distunit; return 6; }
#endif
int lart_clumps_file_read(const char *path, long long *n, double **data, int *present, double *keys, char *distunit) {
    size_t L = strlen(path);
    if ((L > 3 && !strcmp(path + L - 3, ".h5")) || (L > 5 && !strcmp(path + L - 5, ".hdf5"))) return clumps_read_h5(path, n, data, present, keys, distunit);
    return clumps_read_fits(path, n, data, present, keys, distunit);
}


/* ---- reading back one array or keyword of an output file of this host (par%out_merge: write_output_rect.f90:112-420, 780-1000, 1041-1340: the
 * previous file is read, combined photon-weighted with the new run and written again).  section = EXTNAME / group name; name = a table column,
 * or "data" for an image section.  FITS[.gz] or HDF5 by the extension.  0 ok, 1 cannot open, 2 short read, 3 no such section / column /
 * keyword, 4 size mismatch, 6 no HDF5 in this build. */
typedef struct { fits_hdr h; char ttype[64][24]; char extname[72]; const char *want; double keyval; int has_key; } rd_hdr;
static void rd_card(const char *k, const char *v, void *u) {
    rd_hdr *c = (rd_hdr *)u; fits_card(k, v, &c->h);
    if (!strncmp(k, "TTYPE", 5)) { int i = atoi(k + 5); if (i >= 1 && i <= 64) { strncpy(c->ttype[i-1], v, 23); c->ttype[i-1][23] = 0; } }
    else if (!strcmp(k, "EXTNAME")) { strncpy(c->extname, v, 71); c->extname[71] = 0; }
    if (c->want && !strcasecmp(k, c->want)) { c->keyval = atof(v); c->has_key = 1; }       /* FITS keeps 8 characters: the caller passes the truncated name */
}
static int fits_find(gzFile f, const char *section, rd_hdr *c, const char *key) {
    for (int hdu = 0; hdu < 64; hdu++) {
        memset(c, 0, sizeof *c); c->want = key;
        if (fits_header(f, rd_card, c)) return 3;
        if (!strcmp(c->extname, section)) return 0;
        long long skip = 0;
        if (c->h.naxis > 0) { skip = labs(c->h.bitpix) / 8; for (long i = 0; i < c->h.naxis; i++) skip *= c->h.naxes[i]; skip += c->h.pcount; skip = (skip + 2879) / 2880 * 2880; }
        char junk[2880]; for (long long s = 0; s < skip; s += 2880) if (gzread(f, junk, 2880) != 2880) return 3;
    }
    return 3;
}
static int is_h5(const char *path) { size_t L = strlen(path); return (L > 3 && !strcmp(path + L - 3, ".h5")) || (L > 5 && !strcmp(path + L - 5, ".hdf5")); }
int lart_out_read(const char *path, const char *section, const char *name, double *dst, long long n) {
    if (is_h5(path)) {
#ifdef LART_HDF5
        H5E_auto2_t ef; void *ed; H5Eget_auto2(H5E_DEFAULT, &ef, &ed); H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
        hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT); int rc = 1;
        if (f >= 0) { rc = 3;
            if (H5Lexists(f, section, H5P_DEFAULT) > 0) { hid_t g = H5Gopen2(f, section, H5P_DEFAULT);
                if (g >= 0) { if (H5Lexists(g, name, H5P_DEFAULT) > 0) { hid_t ds = H5Dopen2(g, name, H5P_DEFAULT);
                        if (ds >= 0) { hid_t sp = H5Dget_space(ds); hssize_t np = H5Sget_simple_extent_npoints(sp); H5Sclose(sp);
                            rc = (np != (hssize_t)n) ? 4 : (H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, dst) < 0 ? 2 : 0); H5Dclose(ds); } }
                    H5Gclose(g); } }
            H5Fclose(f); }
        H5Eset_auto2(H5E_DEFAULT, ef, ed);
        return rc;
#else
        return 6;
#endif
    }
    gzFile f = gzopen(path, "rb"); if (!f) return 1;
    rd_hdr c; int rc = fits_find(f, section, &c, NULL);
    if (rc) { gzclose(f); return rc; }
    if (!c.h.bintable) {                                                   /* image */
        long long np = 1; for (long i = 0; i < c.h.naxis; i++) np *= c.h.naxes[i];
        int w = (int)(labs(c.h.bitpix) / 8);
        if (c.h.naxis < 1 || np != n || (c.h.bitpix != -64 && c.h.bitpix != -32)) { gzclose(f); return 4; }
        unsigned char buf[8 * 512];
        for (long long i = 0; i < n; ) { long long m = n - i < 512 ? n - i : 512;
            if (gzread(f, buf, (unsigned)(m * w)) != m * w) { gzclose(f); return 2; }
            for (long long j = 0; j < m; j++) dst[i + j] = w == 8 ? be_f64(buf + 8 * j) : be_f32(buf + 4 * j);
            i += m; }
        gzclose(f); return 0;
    }
    int off[64], width[64], kind[64], row = 0, col = -1;
    for (long i = 0; i < c.h.tfields && i < 64; i++) {
        const char *t = c.h.tform[i]; int rep = 0; while (*t >= '0' && *t <= '9') { rep = rep * 10 + (*t - '0'); t++; } if (rep == 0) rep = 1;
        int w = 0, k = 0;
        switch (*t) { case 'D': w = 8; k = 0; break; case 'E': w = 4; k = 1; break; case 'J': w = 4; k = 2; break; case 'K': w = 8; k = 2; break; case 'I': w = 2; k = 2; break; default: gzclose(f); return 4; }
        off[i] = row; width[i] = w; kind[i] = k; row += w * rep;
        if (!strcmp(c.ttype[i], name)) col = (int)i;
    }
    if (col < 0) { gzclose(f); return 3; }
    if (row != c.h.naxes[0] || c.h.naxes[1] != n) { gzclose(f); return 4; }
    unsigned char *buf = (unsigned char *)malloc((size_t)row);
    if (!buf) { gzclose(f); return 5; }
    for (long long r = 0; r < n; r++) {
        if (gzread(f, buf, (unsigned)row) != row) { free(buf); gzclose(f); return 2; }
        const unsigned char *b = buf + off[col]; dst[r] = kind[col] == 0 ? be_f64(b) : kind[col] == 1 ? be_f32(b) : be_int(b, width[col]);
    }
    free(buf); gzclose(f); return 0;
}
int lart_out_read_key(const char *path, const char *section, const char *key, double *v) {
    if (is_h5(path)) {
#ifdef LART_HDF5
        H5E_auto2_t ef; void *ed; H5Eget_auto2(H5E_DEFAULT, &ef, &ed); H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
        hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT); int rc = 1;
        if (f >= 0) { rc = 3;
            if (H5Lexists(f, section, H5P_DEFAULT) > 0) { hid_t g = H5Gopen2(f, section, H5P_DEFAULT); if (g >= 0) { rc = h5_attr_f64(g, key, v) ? 3 : 0; H5Gclose(g); } }
            H5Fclose(f); }
        H5Eset_auto2(H5E_DEFAULT, ef, ed);
        return rc;
#else
        return 6;
#endif
    }
    gzFile f = gzopen(path, "rb"); if (!f) return 1;
    char k8[9]; strncpy(k8, key, 8); k8[8] = 0;
    rd_hdr c; int rc = fits_find(f, section, &c, k8);
    gzclose(f);
    if (rc) return rc;
    if (!c.has_key) return 3;
    *v = c.keyval; return 0;
}
