/* declarations only (tests/support/psrdada_api/README.md); used at src/dada_handler.hh:26-28,70,77,120 */
#ifndef DSABF_TEST_MULTILOG_H
#define DSABF_TEST_MULTILOG_H
#include <stdio.h>
#include <syslog.h>
typedef struct multilog_s multilog_t;
multilog_t *multilog_open(const char *program_name, char syslog);
int multilog_add(multilog_t *m, FILE *fptr);
int multilog(multilog_t *m, int priority, const char *format, ...);
#endif
