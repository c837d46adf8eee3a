/* Test infrastructure: runs the unmodified JM encoder (libjm.so built from /root/reference). */
extern int jm_main(int argc, char **argv);
int main(int argc, char **argv) { return jm_main(argc, argv); }
