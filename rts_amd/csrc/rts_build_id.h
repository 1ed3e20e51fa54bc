#define RTS_SOURCE_HASH "e3e6cb5a5a3ecc32"
