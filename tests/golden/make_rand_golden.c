/* Prints RAND_MAX and the first 4096 values of the C library's rand() in a process that never calls srand():
 * the stream behind Eigen::Vector4d::Random() in the reference's DlsPnp (dls_pnp.cc:134). */
#include <stdio.h>
#include <stdlib.h>
int main(void) {
  printf("%d\n", RAND_MAX);
  for (int i = 0; i < 4096; ++i) printf("%d\n", rand());
  return 0;
}
