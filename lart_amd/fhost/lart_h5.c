/* Output-container helpers of the Fortran host: the HDF5 layout of the reference (hdf5io_mod.f90:8-13: one group per
 * section named by its EXTNAME, image sections hold a dataset `data`, table sections one 1-D dataset per column,
 * keywords are attributes of the group; Fortran array order = reversed HDF5 dimensions) through the HDF5 C API, and gzip
 * for the FITS form (<base>.fits.gz).  Built with -DLART_HDF5 when <hdf5.h> is in the image; otherwise the HDF5 entry
 * points report "unavailable" and the host writes FITS. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
/* fdims = Fortran extents (fastest first); the file gets them reversed, as the HDF5 Fortran API does */
int lh5_put_f64(long long grp, const char *name, int rank, const long long *fdims, const double *a) {
    hsize_t d[8]; if (rank < 1 || rank > 8) return 1;
    for (int i = 0; i < rank; i++) d[i] = (hsize_t)fdims[rank - 1 - i];
    hid_t sp = H5Screate_simple(rank, d, NULL), ds = H5Dcreate2((hid_t)grp, name, H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
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
