/* Declarations-only stand-in for OTP's erl_nif.h, written from the public erl_nif documentation,
 * used by tests/test_abi.py to TYPE-CHECK send-slam_amd/nif/sendslam_nif.c (gcc -fsyntax-only)
 * against include/sendslam_orb.h in a container without Erlang/OTP.  It declares only what the
 * glue uses, defines nothing, and is never linked: a real build uses the real header. */
#ifndef ERL_NIF_DECLS_H
#define ERL_NIF_DECLS_H
#include <stddef.h>

typedef unsigned long ERL_NIF_TERM;
typedef struct enif_environment_t ErlNifEnv;
typedef struct enif_resource_type_t ErlNifResourceType;
typedef struct { size_t size; unsigned char *data; void *ref_bin; void *spare[2]; } ErlNifBinary;
typedef struct { const char *name; unsigned arity; ERL_NIF_TERM (*fptr)(ErlNifEnv *, int, const ERL_NIF_TERM[]); unsigned flags; } ErlNifFunc;
typedef void ErlNifResourceDtor(ErlNifEnv *, void *);
typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;
typedef enum { ERL_NIF_LATIN1 = 1 } ErlNifCharEncoding;
#define ERL_NIF_DIRTY_JOB_CPU_BOUND 1
#define ERL_NIF_DIRTY_JOB_IO_BOUND 2

ErlNifResourceType *enif_open_resource_type(ErlNifEnv *, const char *, const char *, ErlNifResourceDtor *, ErlNifResourceFlags, ErlNifResourceFlags *);
void *enif_alloc_resource(ErlNifResourceType *, size_t);
void enif_release_resource(void *);
ERL_NIF_TERM enif_make_resource(ErlNifEnv *, void *);
int enif_get_resource(ErlNifEnv *, ERL_NIF_TERM, ErlNifResourceType *, void **);
int enif_get_int(ErlNifEnv *, ERL_NIF_TERM, int *);
int enif_get_double(ErlNifEnv *, ERL_NIF_TERM, double *);
int enif_get_tuple(ErlNifEnv *, ERL_NIF_TERM, int *, const ERL_NIF_TERM **);
int enif_inspect_binary(ErlNifEnv *, ERL_NIF_TERM, ErlNifBinary *);
unsigned char *enif_make_new_binary(ErlNifEnv *, size_t, ERL_NIF_TERM *);
void *enif_alloc(size_t);
void enif_free(void *);
ERL_NIF_TERM enif_make_atom(ErlNifEnv *, const char *);
ERL_NIF_TERM enif_make_int(ErlNifEnv *, int);
ERL_NIF_TERM enif_make_double(ErlNifEnv *, double);
ERL_NIF_TERM enif_make_string(ErlNifEnv *, const char *, ErlNifCharEncoding);
ERL_NIF_TERM enif_make_badarg(ErlNifEnv *);
ERL_NIF_TERM enif_make_tuple(ErlNifEnv *, unsigned, ...);
int enif_get_list_length(ErlNifEnv *, ERL_NIF_TERM, unsigned *);
int enif_get_list_cell(ErlNifEnv *, ERL_NIF_TERM, ERL_NIF_TERM *, ERL_NIF_TERM *);
ERL_NIF_TERM enif_make_list_from_array(ErlNifEnv *, const ERL_NIF_TERM[], unsigned);
#define enif_make_tuple2(env, a, b) enif_make_tuple(env, 2, a, b)
#define enif_make_tuple3(env, a, b, c) enif_make_tuple(env, 3, a, b, c)
#define enif_make_tuple4(env, a, b, c, d) enif_make_tuple(env, 4, a, b, c, d)
#define enif_make_tuple5(env, a, b, c, d, e) enif_make_tuple(env, 5, a, b, c, d, e)
#define ERL_NIF_INIT(MODULE, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD) \
    int sendslam_nif_decl_check_(void) { return (int)(sizeof(FUNCS) / sizeof(FUNCS[0])) + ((LOAD) != 0); }
#endif
