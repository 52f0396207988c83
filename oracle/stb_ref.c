/* ORACLE side only: compiles the reference's vendored stb_image.h (v2.28) from where it
 * lies under /root/reference, to validate this repo's own image decoder bit-for-bit.
 * No reference source is copied: the header is included by absolute path at build time. */
#define STB_IMAGE_IMPLEMENTATION
#define STBIDEF __attribute__((visibility("default")))
#include RR_REF_STB_PATH
