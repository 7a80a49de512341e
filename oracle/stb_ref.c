/* oracle/stb_ref.c -- TEST INFRASTRUCTURE: builds the REFERENCE's own PNG decoder (its vendored include/stb_image.h, which
 * src/render/cubemap.cpp:40 calls as stbi_load(path, &w, &h, &n, 0)) into oracle/_ref/libstb_ref.so, straight from the file
 * where it lies under /root/reference -- nothing is copied into this repository.  tests/test_host_parity.py decodes the cube-map
 * crosses with it and compares the product's zlib-based reader (rt_load_png) byte for byte.
 * Recipe: `make -C oracle ref` (needs /root/reference; the built library travels to the GPU box like the other .so files). */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_ONLY_PNG
#include "stb_image.h"
