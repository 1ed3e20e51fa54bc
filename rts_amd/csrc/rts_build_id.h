#define RTS_SOURCE_HASH "12ab77bb8bc9fee4"
