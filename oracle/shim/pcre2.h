/*
 * TEST INFRASTRUCTURE ONLY (oracle/): declaration-only stand-in for <pcre2.h>.
 *
 * The image ships the PCRE2 runtime (libpcre2-8.so.0, 10.39, Unicode 14.0.0, JIT) but not the
 * development header.  The reference core (/root/reference/src/tiktoken/tiktoken.hpp:7-8) includes
 * <pcre2.h>; this file declares exactly the subset of the public PCRE2 C API that the reference
 * calls (tiktoken.cpp:21,32,37,51-58,63,87-93,107; tiktoken.hpp:71) so the unmodified reference
 * sources compile where they lie.  Constants are the values of the public PCRE2 10.x API.
 * It also pulls in <limits.h>/<stdio.h>/<stdlib.h>, which the reference uses (INT_MAX, printf,
 * atexit) without including.
 */
#ifndef TD_ORACLE_PCRE2_SHIM_H
#define TD_ORACLE_PCRE2_SHIM_H

#include <limits.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t PCRE2_UCHAR8;
typedef const PCRE2_UCHAR8 *PCRE2_SPTR8;
typedef size_t PCRE2_SIZE;

typedef struct pcre2_real_code_8 pcre2_code_8;
typedef struct pcre2_real_match_data_8 pcre2_match_data_8;
typedef struct pcre2_real_general_context_8 pcre2_general_context_8;
typedef struct pcre2_real_compile_context_8 pcre2_compile_context_8;
typedef struct pcre2_real_match_context_8 pcre2_match_context_8;

/* compile options */
#define PCRE2_UCP 0x00020000u
#define PCRE2_UTF 0x00080000u
/* match options */
#define PCRE2_NOTEMPTY 0x00000004u
#define PCRE2_NO_UTF_CHECK 0x40000000u
/* misc */
#define PCRE2_ZERO_TERMINATED (~(PCRE2_SIZE)0)
#define PCRE2_ERROR_NOMATCH (-1)
#define PCRE2_JIT_COMPLETE 0x00000001u
#define PCRE2_CONFIG_UNICODE_VERSION 10
#define PCRE2_CONFIG_VERSION 11

pcre2_code_8 *pcre2_compile_8(PCRE2_SPTR8 pattern, PCRE2_SIZE length, uint32_t options,
                              int *errorcode, PCRE2_SIZE *erroroffset,
                              pcre2_compile_context_8 *ccontext);
void pcre2_code_free_8(pcre2_code_8 *code);
int pcre2_jit_compile_8(pcre2_code_8 *code, uint32_t options);
int pcre2_match_8(const pcre2_code_8 *code, PCRE2_SPTR8 subject, PCRE2_SIZE length,
                  PCRE2_SIZE startoffset, uint32_t options, pcre2_match_data_8 *match_data,
                  pcre2_match_context_8 *mcontext);
pcre2_match_data_8 *pcre2_match_data_create_from_pattern_8(const pcre2_code_8 *code,
                                                           pcre2_general_context_8 *gcontext);
void pcre2_match_data_free_8(pcre2_match_data_8 *match_data);
PCRE2_SIZE *pcre2_get_ovector_pointer_8(pcre2_match_data_8 *match_data);
int pcre2_config_8(uint32_t what, void *where);

/* generic names, as <pcre2.h> provides them for PCRE2_CODE_UNIT_WIDTH == 8 */
#define pcre2_code pcre2_code_8
#define pcre2_match_data pcre2_match_data_8
#define pcre2_compile pcre2_compile_8
#define pcre2_code_free pcre2_code_free_8
#define pcre2_jit_compile pcre2_jit_compile_8
#define pcre2_match pcre2_match_8
#define pcre2_match_data_create_from_pattern pcre2_match_data_create_from_pattern_8
#define pcre2_match_data_free pcre2_match_data_free_8
#define pcre2_get_ovector_pointer pcre2_get_ovector_pointer_8
#define pcre2_config pcre2_config_8

#ifdef __cplusplus
}
#endif
#endif
