// spl_error.h -- error reporting shared by the translation units of libspliser_hip.so.
#ifndef SPL_ERROR_H
#define SPL_ERROR_H
// Records a printf-style message for spl_last_error() on this thread and returns `code`.
int spl_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#endif
