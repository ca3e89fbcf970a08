/* Writes tests/golden/keras_weights_sample.h5 with the REAL HDF5 library (libhdf5 1.10, default "earliest" file format --
 * what h5py / Keras 2.1 produce): the layout keras.engine.topology.save_weights_to_hdf5_group writes
 *   /            attrs layer_names (fixed-length byte strings), backend, keras_version (variable-length strings)
 *   /<layer>     attr  weight_names;   /<layer>/<weight name>  float32 datasets (weight names contain '/': nested groups)
 * 13 layers (more than one symbol-table node: the root B-tree has several leaves), a layer without weights, a nested
 * (TimeDistributed / sub-model) layer holding two inner layers, one chunked dataset.  Values: v[i] = ((7 i + 3 L) mod 101)/8 - 5
 * for the L-th dataset written, so the test regenerates them.
 * Build + run (build container; the library lives in /opt/conda):
 *   gcc tests/golden/make_h5_fixture.c -I/opt/conda/include -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib -o /tmp/mkh5 && /tmp/mkh5 tests/golden/keras_weights_sample.h5
 */
#include "hdf5.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int L = 0;

static void str_array_attr(hid_t obj, const char* name, const char** vals, int n) {
    size_t w = 1;
    for (int i = 0; i < n; ++i) if (strlen(vals[i]) > w) w = strlen(vals[i]);
    char* buf = calloc((size_t)(n ? n : 1), w);
    for (int i = 0; i < n; ++i) memcpy(buf + i * w, vals[i], strlen(vals[i]));     /* null-padded like numpy 'S' */
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, w);
    H5Tset_strpad(t, H5T_STR_NULLPAD);
    hsize_t d = (hsize_t)n;
    hid_t s = H5Screate_simple(1, &d, NULL);
    hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    if (n) H5Awrite(a, t, buf);
    H5Aclose(a); H5Sclose(s); H5Tclose(t); free(buf);
}

static void vlen_str_attr(hid_t obj, const char* name, const char* val) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, &val);
    H5Aclose(a); H5Sclose(s); H5Tclose(t);
}

static void dataset(hid_t layer, const char* path, int rank, const hsize_t* dims, int chunked) {
    size_t n = 1;
    for (int i = 0; i < rank; ++i) n *= dims[i];
    float* v = malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) v[i] = (float)((7 * i + 3 * (size_t)L) % 101) / 8.0f - 5.0f;
    ++L;
    hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(lcpl, 1);                 /* 'conv1/kernel:0' creates the inner group, as h5py does */
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    if (chunked) { hsize_t c[4] = {2, 2, 2, 2}; for (int i = 0; i < rank; ++i) if (c[i] > dims[i]) c[i] = dims[i]; H5Pset_chunk(dcpl, rank, c); }
    hid_t s = H5Screate_simple(rank, dims, NULL);
    hid_t d = H5Dcreate2(layer, path, H5T_IEEE_F32LE, s, lcpl, dcpl, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v);
    H5Dclose(d); H5Sclose(s); H5Pclose(dcpl); H5Pclose(lcpl); free(v);
}

int main(int argc, char** argv) {
    const char* out = argc > 1 ? argv[1] : "keras_weights_sample.h5";
    hid_t f = H5Fcreate(out, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    const char* layers[13] = {"input_1", "conv1", "bn_conv1", "res2a_branch2a", "bn2a_branch2a", "res2a_branch2b", "bn2a_branch2b",
                              "fpn_c5p5", "rpn_conv_shared", "mrcnn_class_conv1", "mrcnn_class_bn1", "imgcap_caption_td", "imgcap_embedding_layer"};
    str_array_attr(f, "layer_names", layers, 13);
    vlen_str_attr(f, "backend", "tensorflow");
    vlen_str_attr(f, "keras_version", "2.1.6");
    for (int i = 0; i < 13; ++i) {
        hid_t g = H5Gcreate2(f, layers[i], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        char a[64], b[64], c[64], d[64];
        if (i == 0) {
            str_array_attr(g, "weight_names", NULL, 0);
        } else if (strncmp(layers[i], "bn", 2) == 0 || strstr(layers[i], "_bn")) {
            snprintf(a, 64, "%s/gamma:0", layers[i]); snprintf(b, 64, "%s/beta:0", layers[i]);
            snprintf(c, 64, "%s/moving_mean:0", layers[i]); snprintf(d, 64, "%s/moving_variance:0", layers[i]);
            const char* w[4] = {a, b, c, d};
            str_array_attr(g, "weight_names", w, 4);
            hsize_t dim = 6;
            for (int k = 0; k < 4; ++k) dataset(g, w[k], 1, &dim, 0);
        } else if (strcmp(layers[i], "imgcap_caption_td") == 0) {         /* a nested model: two inner layers in one layer group */
            const char* w[5] = {"imgcap_lstm1/kernel:0", "imgcap_lstm1/recurrent_kernel:0", "imgcap_lstm1/bias:0", "imgcap_lstm_d2/kernel:0",
                                "imgcap_lstm_d2/bias:0"};
            str_array_attr(g, "weight_names", w, 5);
            hsize_t k1[2] = {10, 16}, k2[2] = {4, 16}, b1 = 16, k3[2] = {8, 12}, b3 = 12;
            dataset(g, w[0], 2, k1, 0); dataset(g, w[1], 2, k2, 0); dataset(g, w[2], 1, &b1, 0); dataset(g, w[3], 2, k3, 0); dataset(g, w[4], 1, &b3, 0);
        } else if (strcmp(layers[i], "imgcap_embedding_layer") == 0) {
            const char* w[1] = {"imgcap_embedding_layer/embeddings:0"};
            str_array_attr(g, "weight_names", w, 1);
            hsize_t e[2] = {9, 5};
            dataset(g, w[0], 2, e, 1);                                     /* chunked, unfiltered */
        } else {
            snprintf(a, 64, "%s/kernel:0", layers[i]); snprintf(b, 64, "%s/bias:0", layers[i]);
            const char* w[2] = {a, b};
            str_array_attr(g, "weight_names", w, 2);
            hsize_t k[4] = {3, 3, 2, 6}, bias = 6;
            dataset(g, w[0], 4, k, 0); dataset(g, w[1], 1, &bias, 0);
        }
        H5Gclose(g);
    }
    H5Fclose(f);
    printf("%s: %d datasets\n", out, L);
    return 0;
}
