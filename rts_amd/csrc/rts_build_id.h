#define RTS_SOURCE_HASH "a1ce3adce15b5d92"
