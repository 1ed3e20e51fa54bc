#define RTS_SOURCE_HASH "aafcabcf5ea5198e"
