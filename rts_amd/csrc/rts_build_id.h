#define RTS_SOURCE_HASH "232ce2d03e6e1cce"
