// stand-ins for the two library symbols child_q4.hip refers to, so that ONE translation unit builds into an experiment library
#include <cstdarg>
#include <cstdio>
int g_child_nw = 0, g_child_depth = 0;
void pcgc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
