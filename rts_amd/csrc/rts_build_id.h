#define RTS_SOURCE_HASH "7e7efb2ebaaf52d4"
