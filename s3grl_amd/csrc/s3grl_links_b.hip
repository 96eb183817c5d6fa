// link kernels of sign_k = 5, 6: see the end of s3grl_structure.hip
#define S3GRL_LINKS_PART s3grl_links_part_b
#define S3GRL_LINKS_K0 5
#define S3GRL_LINKS_K1 6
#define S3GRL_TOUCH_UNIT links_b
#include "s3grl_structure.hip"
