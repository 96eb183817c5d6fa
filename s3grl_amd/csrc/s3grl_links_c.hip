// link kernels of sign_k = 7, 8: see the end of s3grl_structure.hip
#define S3GRL_LINKS_PART s3grl_links_part_c
#define S3GRL_LINKS_K0 7
#define S3GRL_LINKS_K1 8
#define S3GRL_TOUCH_UNIT links_c
#include "s3grl_structure.hip"
