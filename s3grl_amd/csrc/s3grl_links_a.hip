// link kernels of sign_k = 1, 2: see the end of s3grl_structure.hip
#define S3GRL_LINKS_PART s3grl_links_part_a
#define S3GRL_LINKS_K0 1
#define S3GRL_LINKS_K1 2
#define S3GRL_TOUCH_UNIT links_a
#include "s3grl_structure.hip"
