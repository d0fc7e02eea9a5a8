"""Error convention of the reference's safe layer (S/ZstdException.cs, S/ThrowHelper.cs:7-41, U/ZSTD_ErrorCode.cs)."""
import enum


class ZSTD_ErrorCode(enum.IntEnum):
    ZSTD_error_no_error = 0
    ZSTD_error_GENERIC = 1
    ZSTD_error_prefix_unknown = 10
    ZSTD_error_version_unsupported = 12
    ZSTD_error_frameParameter_unsupported = 14
    ZSTD_error_frameParameter_windowTooLarge = 16
    ZSTD_error_corruption_detected = 20
    ZSTD_error_checksum_wrong = 22
    ZSTD_error_dictionary_corrupted = 30
    ZSTD_error_dictionary_wrong = 32
    ZSTD_error_dictionaryCreation_failed = 34
    ZSTD_error_parameter_unsupported = 40
    ZSTD_error_parameter_outOfBound = 42
    ZSTD_error_tableLog_tooLarge = 44
    ZSTD_error_maxSymbolValue_tooLarge = 46
    ZSTD_error_maxSymbolValue_tooSmall = 48
    ZSTD_error_stage_wrong = 60
    ZSTD_error_init_missing = 62
    ZSTD_error_memory_allocation = 64
    ZSTD_error_workSpace_tooSmall = 66
    ZSTD_error_dstSize_tooSmall = 70
    ZSTD_error_srcSize_wrong = 72
    ZSTD_error_dstBuffer_null = 74
    ZSTD_error_frameIndex_tooLarge = 100
    ZSTD_error_seekableIO = 102
    ZSTD_error_dstBuffer_wrong = 104
    ZSTD_error_srcBuffer_wrong = 105
    ZSTD_error_maxCode = 120


SIZE_MAX = (1 << 64) - 1
CONTENTSIZE_UNKNOWN = SIZE_MAX          # (0ULL - 1), S/ThrowHelper.cs:7
CONTENTSIZE_ERROR = SIZE_MAX - 1        # (0ULL - 2), S/ThrowHelper.cs:8
DST_SIZE_TOO_SMALL = SIZE_MAX - 70 + 1  # (size_t)-70, the value TryWrap/TryUnwrap compare against


class ZstdException(Exception):
    """S/ZstdException.cs: carries the ZSTD_ErrorCode next to the message."""

    def __init__(self, code, message):
        super().__init__(message)
        self.Code = self.code = ZSTD_ErrorCode(code) if code in ZSTD_ErrorCode._value2member_map_ else code


def is_error(value: int) -> bool:
    return value > (1 << 64) - 120          # U/ErrorPrivate.cs:10-13


def get_error_code(value: int) -> int:
    return 0 if not is_error(value) else (1 << 64) - value


def ensure_zstd_success(lib, value: int) -> int:
    """S/ThrowHelper.cs:10-24."""
    if is_error(value):
        raise ZstdException(get_error_code(value), lib.ZSTD_getErrorName(value).decode())
    return value


def ensure_content_size_ok(size: int) -> int:
    """S/ThrowHelper.cs:26-35."""
    if size == CONTENTSIZE_UNKNOWN:
        raise ZstdException(ZSTD_ErrorCode.ZSTD_error_GENERIC, "Decompressed content size is not specified")
    if size == CONTENTSIZE_ERROR:
        raise ZstdException(ZSTD_ErrorCode.ZSTD_error_GENERIC, "Decompressed content size cannot be determined (e.g. invalid magic number, srcSize too small)")
    return size
